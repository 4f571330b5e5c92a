#!/bin/bash
# One process per GPU of a command line of this package, RCCL transport of libamplisolve_hip.so (no Python):
#   tools/launch_native.sh N AmpliSolveErrorEstimation panel_design=... (7 tokens)
#   tools/launch_native.sh N AmpliSolveVariantCalling errorFile=... (5 tokens)
# Rank k runs on device k (mod the visible ones) unless AMPLISOLVE_DEVICE says otherwise; the communicator id travels through a fresh file next to the output directory.  Exit status:
# non-zero when any shard failed.
set -u
N=$1; shift
EXE=$1; shift
BIN="$(cd "$(dirname "$0")/.." && pwd)/amplisolve_amd/bin"
ID="${TMPDIR:-/tmp}/amplisolve_rccl_id.$$.$(date +%s)"
pids=()
for ((k = 0; k < N; k++)); do
  AMPLISOLVE_WORLD_SIZE=$N AMPLISOLVE_RANK=$k AMPLISOLVE_ID_FILE=$ID "$BIN/$EXE" "$@" > >(sed "s/^/[$k] /") 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait "$p" || rc=1; done
rm -f "$ID"
exit $rc
