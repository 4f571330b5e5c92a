#!/bin/bash
# A/B an environment knob on the GPU box: tools/experiments/probes/ab_env.sh VAR   (runs bench with VAR unset / VAR=1, interleaved)
cd "$(dirname "$0")/../../.."
for round in 1 2 3; do for v in "" "1"; do
  if [ -z "$v" ]; then unset $1; else export $1=1; fi
  python bench.py --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); k=d['kernels']; print('$1=$v', round(k['error_reduce_ms'],4), round(k['poisson_call_ms'],4), round(d['ms_per_step'],4))"
done; done
