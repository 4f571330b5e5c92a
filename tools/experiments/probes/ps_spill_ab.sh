#!/bin/bash
# ON THE GPU BOX: poisson_stream at 8 waves per SIMD with spills (tree) against 7 / 6 waves without (variants ps_lb7 / ps_lb6):
# times in the alternating loop (tools/reduce_bench.py, u16 + u24), then WRITE_SIZE / FETCH_SIZE of the u16 launches
cd "$(dirname "$0")/../../.."
R=$PWD
for round in 1 2; do
  RB_TAG="tree  " python tools/reduce_bench.py u16 u24
  for v in ps_lb7 ps_lb6; do RB_TAG="$v" AMPLISOLVE_HIP_LIB=$R/_variants/$v.so python tools/reduce_bench.py u16 u24; done
done
OUT=$R/gpurun_out/r4b/pspmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in tree ps_lb7; do
  [ $v = tree ] && unset AMPLISOLVE_HIP_LIB || export AMPLISOLVE_HIP_LIB=$R/_variants/$v.so
  PP_LAYOUT=u16 timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE FETCH_SIZE --kernel-trace --output-format csv -d $OUT/$v -- python3 $R/tools/experiments/probes/poisson_prof.py 0 0 20 > $OUT/$v.log 2>&1 || exit 1
done
python3 - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for sub in ("tree", "ps_lb7"):
    for f in glob.glob(f"{out}/{sub}/*/*counter_collection.csv"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if "poisson" in k:
                print(sub, k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
