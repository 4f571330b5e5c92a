#!/bin/bash
# ON THE GPU BOX: kernel-trace stats of a few poisson_call launch shapes + FETCH/WRITE PMC of one of them
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../../.." && pwd)}
OUT=$R/gpurun_out/r2/pprof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "$@"; do
  set -- $cfg
  tag=$(echo $cfg | tr ' ' '_')
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t_$tag -- python3 $R/tools/experiments/probes/poisson_prof.py $cfg > $OUT/t_$tag.log 2>&1 || exit 1
  f=$(find $OUT/t_$tag -name '*kernel_stats.csv' | head -1)
  echo "== $cfg"; tail -1 $OUT/t_$tag.log; grep -E "poisson|Name" $f | cut -d, -f1-5
done
