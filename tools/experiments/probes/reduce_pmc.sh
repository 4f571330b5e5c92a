#!/bin/bash
# ON THE GPU BOX: SQ counters of error_reduce (tools/reduce_bench.py) for the library in AMPLISOLVE_HIP_LIB
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
OUT=$R/gpurun_out/r2/rpmc_$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/tools/reduce_bench.py > $OUT/sq.log 2>&1 || exit 1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq2 -- python3 $R/tools/reduce_bench.py > $OUT/sq2.log 2>&1 || exit 1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("sq", "sq2"):
    f = glob.glob(f"{out}/{sub}/*/*counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        if "error_reduce" in k:
            print(sub, k, {c: round(sum(v) / len(v) / 1e6, 2) for c, v in d.items()}, "(millions) n=", len(next(iter(d.values()))))
PY
