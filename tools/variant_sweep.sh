#!/bin/bash
# Build error_reduce variants into _variants/ (here, no GPU needed) -- then run tools/variant_bench.sh on the GPU box.
set -e
cd "$(dirname "$0")/.."
mkdir -p _variants
for U in 1 2 4; do for PP in 0 1; do
  out=_variants/lib_u${U}_pp${PP}.so
  hipcc --offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -DAMPLI_RED_UNROLL=$U -DAMPLI_RED_PINGPONG=$PP \
    -Rpass-analysis=kernel-resource-usage -o $out amplisolve_amd/csrc/ampli_kernels.hip 2>&1 | grep -A2 "error_reduce_kernelILb1" | grep -o "VGPRs: [0-9]*" | sed "s/^/u$U pp$PP /"
done; done
