#!/bin/bash
cd "$(dirname "$0")/.."
for rows in 1 2 3 4 6 1 2 3 4 6; do
  AMPLI_EXP_ROWS=$rows python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('rows $rows', round(d['kernels']['poisson_call_ms'],4), round(d['ms_per_step'],4))"
done
