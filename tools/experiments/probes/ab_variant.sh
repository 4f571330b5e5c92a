#!/bin/bash
# A/B on one GPU box: the working tree's library against _variants/$1.so, tools/reduce_bench.py, interleaved, two rounds
cd "$(dirname "$0")/../../.."
for round in 1 2; do
  RB_TAG="tree" python tools/reduce_bench.py u16 u24 i32
  RB_TAG="$1" AMPLISOLVE_HIP_LIB=$PWD/_variants/$1.so python tools/reduce_bench.py u16 u24 i32
done
