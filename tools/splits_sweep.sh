#!/bin/bash
cd "$(dirname "$0")/.."
for sp in 1 2 4 1 2 4; do
  python bench.py --steps 30 --warmup 3 --no-cpu-baseline --splits $sp 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('splits $sp', round(d['kernels']['error_reduce_ms'],4), round(d['kernels']['poisson_call_ms'],4), round(d['ms_per_step'],4))"
done
