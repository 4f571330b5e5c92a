#!/bin/bash
set -e
cd "$(dirname "$0")/.."
rm -rf _variants; mkdir -p _variants
for W in 4 5 6 8; do for PF in 1 2 3; do
  out=_variants/lib_w${W}_pf${PF}.so
  hipcc --offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -DAMPLI_PCQ_WAVES_PER_EU=$W -DAMPLI_PCQ_PREFETCH=$PF \
    -Rpass-analysis=kernel-resource-usage -o $out amplisolve_amd/csrc/ampli_kernels.hip 2>&1 | grep -A4 "poisson_call_queue" | grep -oE "VGPRs: [0-9]+|ScratchSize \[bytes/lane\]: [0-9]+" | tr '\n' ' ' | sed "s/^/w$W pf$PF /"; echo
done; done
