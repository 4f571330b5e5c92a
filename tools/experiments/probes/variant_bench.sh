#!/bin/bash
# On the GPU box: bench every lib in _variants/ (interleaved, 2 rounds) and print error_reduce ms
cd "$(dirname "$0")/../../.."
for round in 1 2; do for f in _variants/*.so; do
  AMPLISOLVE_HIP_LIB=$PWD/$f python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$f', round(d['kernels']['error_reduce_ms'],4), round(d['kernels']['poisson_call_ms'],4), round(d['ms_per_step'],4))"
done; done
