#!/bin/bash
# runs on the GPU box: strace of the probe's start-up, several times, to see which system calls a slow hipInit spends its time in
cd "$(dirname "$0")"
OUT=${1:-/tmp}
which strace || { echo "no strace"; exit 0; }
[ -x ./init_probe ] || hipcc -O2 --offload-arch=gfx950 -o init_probe init_probe.cpp -lpthread
for i in 1 2 3 4 5 6 7 8; do
  strace -f -tt -T -o $OUT/strace_$i.txt ./init_probe > $OUT/probe_$i.txt 2>&1
  grep hipInit $OUT/probe_$i.txt
done
