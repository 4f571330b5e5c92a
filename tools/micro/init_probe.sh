#!/bin/bash
# runs on the GPU box: the probe in sequences that show what a process's predecessor does to its start-up
set -e
cd "$(dirname "$0")"
hipcc -O2 --offload-arch=gfx950 -o init_probe init_probe.cpp -lpthread
t() { local s=$(date +%s%N); "$@" > /tmp/probe.out; local e=$(date +%s%N); echo "== wall $(( (e - s) / 1000000 )) ms : $*"; grep -E "hipInit|total in main" /tmp/probe.out; }
echo "### cold"; t ./init_probe; cat /tmp/probe.out
echo "### back to back x4 (small)"; for i in 1 2 3 4; do t ./init_probe; done
echo "### after a process that held 800 MB pinned + 800 MB device and freed them"; ./init_probe 800 > /dev/null; t ./init_probe
echo "### after a process that held 800 MB and exited with _exit (no frees)"; ./init_probe 800 leak > /dev/null; t ./init_probe
echo "### the same, 0.5 s later"; ./init_probe 800 leak > /dev/null; sleep 0.5; t ./init_probe
echo "### exit cost: wall of a holder that frees vs one that _exits"; t ./init_probe 800; t ./init_probe 800 leak
