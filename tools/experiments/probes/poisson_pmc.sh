#!/bin/bash
# ON THE GPU BOX: FETCH_SIZE / WRITE_SIZE / SQ counters of one poisson_call launch shape (separate passes)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
OUT=$R/gpurun_out/r2/ppmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CFG="${1:-4 0 32}"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch -- python3 $R/tools/experiments/probes/poisson_prof.py $CFG 20 > $OUT/fetch.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write -- python3 $R/tools/experiments/probes/poisson_prof.py $CFG 20 > $OUT/write.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/tools/experiments/probes/poisson_prof.py $CFG 20 > $OUT/sq.log 2>&1 || exit 1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("fetch", "write", "sq"):
    f = glob.glob(f"{out}/{sub}/*/*counter_collection.csv")
    if not f:
        print(sub, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"][:48]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "poisson" in k or "error_reduce" in k:
            print(sub, k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
