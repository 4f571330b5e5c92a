#!/bin/bash
# Per-kernel register / scratch / LDS figures of the gfx950 code object of one .hip file (cross-compiles; no GPU needed).
# usage: tools/kstat.sh [file.hip] [name filter]     e.g. tools/kstat.sh amplisolve_amd/csrc/ampli_kernels.hip error_reduce
set -e
SRC=${1:-amplisolve_amd/csrc/ampli_kernels.hip}
FILTER=${2:-.}
D=$(mktemp -d /tmp/kstat.XXXX)
SRC=$(readlink -f "$SRC")
( cd "$D" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off ${KSTAT_FLAGS} --save-temps -o k.so "$SRC" )
python3 - "$D" "$FILTER" <<'PY'
import re, sys, glob, subprocess
s = open(glob.glob(sys.argv[1] + "/*gfx950.s")[0]).read()
filt = re.compile(sys.argv[2])
for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", s, re.S):
    blk = m.group(0)
    g = lambda k: re.search(r"\." + k + r":\s+(\S+)", blk).group(1)
    name = g("name")
    if not filt.search(name):
        continue
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip().split("(")[0]
    print(f"{dem:70s} vgpr {g('vgpr_count'):>4s} sgpr {g('sgpr_count'):>4s} scratch {g('private_segment_fixed_size'):>5s} lds {g('group_segment_fixed_size'):>6s}")
PY
echo "asm: $D"
