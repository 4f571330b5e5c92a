#!/bin/bash
# ON THE GPU BOX: error_reduce one wave per position (shipped) against the role-split kernel (two waves per position, uint16 records),
# tools/reduce_bench.py interleaved; then occupancy / VALU counters of both
cd "$(dirname "$0")/../../.."
R=$PWD
for round in 1 2 3; do
  RB_TAG="one-wave  " python tools/reduce_bench.py u16
  RB_TAG="role-split" RB_ROLE_SPLIT=1 python tools/reduce_bench.py u16
done
OUT=$R/gpurun_out/r4b/splitpmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in one split; do
  [ $v = split ] && export RB_ROLE_SPLIT=1 || unset RB_ROLE_SPLIT
  timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/sq_$v -- python3 $R/tools/reduce_bench.py u16 > $OUT/sq_$v.log 2>&1 || echo "sq pass failed $v"
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch_$v -- python3 $R/tools/reduce_bench.py u16 > $OUT/fetch_$v.log 2>&1 || echo "fetch pass failed $v"
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("sq_one", "sq_split", "fetch_one", "fetch_split"):
    for f in glob.glob(f"{out}/{sub}/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:44]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if "error_reduce" in k:
                print(sub, k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
