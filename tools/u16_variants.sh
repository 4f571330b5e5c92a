#!/bin/bash
# On the GPU box: tools/u16_bench.py with every lib in _variants/ (diagnostic builds: timings only)
cd "$(dirname "$0")/.."
for f in "" _variants/*.so; do
  echo "== ${f:-shipped}"
  if [ -n "$f" ]; then export AMPLISOLVE_HIP_LIB=$PWD/$f; else unset AMPLISOLVE_HIP_LIB; fi
  python tools/u16_bench.py 2>/dev/null | grep -v amdgpu | tail -3
done
