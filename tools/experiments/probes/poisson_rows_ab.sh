#!/bin/bash
# ON THE GPU BOX: poisson_call launch shape on config 3 with uint16 records (tools/reduce_bench.py's alternating loop):
# rows per wave x drain workgroups per shard, interleaved, two rounds.  0 = the library's choice (24 rows, 32 workgroups here).
# usage: poisson_rows_ab.sh ["rows list"] ["drain list"]
cd "$(dirname "$0")/../../.."
ROWS=${1:-"0 24 12 8 6 4"}
DRAIN=${2:-"0 8"}
for round in 1 2; do
  for rows in $ROWS; do
    for drain in $DRAIN; do
      RB_TAG="rows=$rows drain=$drain" RB_ROWS=$rows RB_DRAIN=$drain python tools/reduce_bench.py u16
    done
  done
done 2>&1 | grep -v amdgpu.ids
