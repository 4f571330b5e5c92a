#!/bin/bash
# ON THE GPU BOX: occupancy / VALU / fetch counters of error_reduce on uint16 records: general kernel against the compact-state one
# usage: compact_pmc.sh [general|compact|<name of _variants/<name>.so> ...]   (default: general compact)   PASSES="sq mix fetch"
cd "$(dirname "$0")/../../.."
R=$PWD
OUT=$R/gpurun_out/${PMC_DIR:-r4d}/compactpmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
VARIANTS="${@:-general compact}"
PASSES="${PASSES:-sq mix fetch}"
for v in $VARIANTS; do
  unset AMPLISOLVE_HIP_LIB
  export RB_COMPACT=1
  [ $v = general ] && export RB_COMPACT=0
  [ $v != general ] && [ $v != compact ] && export AMPLISOLVE_HIP_LIB=$R/_variants/$v.so
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_$v -- python3 $R/tools/reduce_bench.py u16 > $OUT/trace_$v.log 2>&1 || echo "trace failed $v"
  [[ " $PASSES " == *" sq "* ]] && timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq_$v -- python3 $R/tools/reduce_bench.py u16 > $OUT/sq_$v.log 2>&1 || echo "sq pass failed $v"
  [[ " $PASSES " == *" mix "* ]] && timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_INSTS_LDS --kernel-trace --output-format csv -d $OUT/mix_$v -- python3 $R/tools/reduce_bench.py u16 > $OUT/mix_$v.log 2>&1 || echo "mix pass failed $v"
  [[ " $PASSES " == *" fetch "* ]] && timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$v -- python3 $R/tools/reduce_bench.py u16 > $OUT/fetch_$v.log 2>&1 || echo "fetch pass failed $v"
done
python3 - $OUT $VARIANTS <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for v in sys.argv[2:]:
    for f in glob.glob(f"{out}/trace_{v}/*/*kernel_stats.csv"):
        for r in csv.DictReader(open(f)):
            if "error_reduce" in r["Name"]:
                print(v, "rocprofv3 --stats", r["Name"][:36], "calls", r["Calls"], "avg ns", r["AverageNs"], "min", r["MinNs"], "max", r["MaxNs"])
    for sub in ("sq", "mix", "fetch"):
        for f in glob.glob(f"{out}/{sub}_{v}/*/*counter_collection.csv"):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                acc[r["Kernel_Name"][:36]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, d in acc.items():
                if "error_reduce" in k:
                    print(v, sub, k, {c: round(sum(x) / len(x), 1) for c, x in d.items()}, "n=", len(next(iter(d.values()))))
PY
