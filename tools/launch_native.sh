#!/bin/bash
# One process per GPU of a command line of this package, RCCL transport of libamplisolve_hip.so (no Python):
#   tools/launch_native.sh N AmpliSolveErrorEstimation panel_design=... (7 tokens)
#   tools/launch_native.sh N AmpliSolveVariantCalling errorFile=... (5 tokens)
# Rank k runs on device k (mod the visible ones) unless AMPLISOLVE_DEVICE says otherwise; the communicator id travels through a
# fresh file (per-launch name and AMPLISOLVE_JOB_NONCE).  The FIRST shard that fails ends the job: RCCL has no timeout, so the
# other ranks would sit in their next collective for ever -- they are sent SIGTERM (then SIGKILL) and the launch returns 1.
set -u
N=$1; shift
EXE=$1; shift
BIN="$(cd "$(dirname "$0")/.." && pwd)/amplisolve_amd/bin"
NONCE="$$$(date +%s)"
ID="${TMPDIR:-/tmp}/amplisolve_rccl_id.$NONCE"
pids=()
for ((k = 0; k < N; k++)); do
  AMPLISOLVE_WORLD_SIZE=$N AMPLISOLVE_RANK=$k AMPLISOLVE_ID_FILE=$ID AMPLISOLVE_JOB_NONCE=$NONCE "$BIN/$EXE" "$@" > >(sed "s/^/[$k] /") 2>&1 &
  pids+=($!)
done
rc=0
left=$N
while ((left > 0)); do
  wait -n "${pids[@]}" 2>/dev/null
  st=$?
  ((st == 127)) && break   # nothing left to wait for
  left=$((left - 1))
  if ((st != 0)); then
    rc=1
    echo "launch_native: a shard ended with status $st; stopping the others" >&2
    for p in "${pids[@]}"; do kill -TERM "$p" 2>/dev/null; done
    sleep 2
    for p in "${pids[@]}"; do kill -KILL "$p" 2>/dev/null; done
    break
  fi
done
wait 2>/dev/null
rm -f "$ID"
exit $rc
