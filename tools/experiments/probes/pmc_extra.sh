#!/bin/bash
# extra SQ counters of the default bench command (kernel legs only): instruction mix and issue stalls
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
OUT=$R/gpurun_out/pmc_extra
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --sustained 0 --cold-batches 0 --whole-rounds 0 --no-split-ranges"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA --kernel-trace --output-format csv -d $OUT/a -- $CMD > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; }
timeout -k 10 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_IFETCH SQ_INST_LEVEL_VMEM SQ_WAVES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/b -- $CMD > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; }
python3 - <<PY
import csv, glob, collections
pm = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "error_reduce_kernel<true, 1, 1>" in k or "poisson_stream_kernel<1" in k:
            pm[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in pm.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:24s} {sum(v)/len(v):14.0f}   ({len(v)} dispatches)")
PY
