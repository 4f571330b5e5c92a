#!/bin/bash
# ON THE GPU BOX: variants of the compact uint16 error_reduce kernel (_variants/c_*.so), tools/reduce_bench.py interleaved
cd "$(dirname "$0")/../../.."
R=$PWD
for round in 1 2 3; do
  for v in "$@"; do RB_TAG="$v" AMPLISOLVE_HIP_LIB=$R/_variants/$v.so python tools/reduce_bench.py u16; done
done 2>&1 | grep -v amdgpu.ids
