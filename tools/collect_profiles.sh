#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel-trace stats + PMC passes of the default bench command (kernels only:
# the CPU baseline and end-to-end legs are switched off).  Outputs under gpurun_out/prof_<round>/ (ROUND, default r05) (merged back by gpurun);
# tools/make_profile_summary.py turns them into the files committed under profiles/<round>/.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out/prof_${ROUND:-r06}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --sustained 0 --cold-batches 0 --whole-rounds 0 --no-shard-projection --no-split-ranges"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $CMD > $OUT/bench_under_trace.log 2>&1 || exit 1
# the same with the position ranges of the default command (two launches per kernel and pass, overlapping on two streams)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_ranges -- ${CMD/ --no-split-ranges/} > $OUT/bench_under_trace_ranges.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/pmc_fetch -- $CMD > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_write -- $CMD > $OUT/pmc_write.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_sq -- $CMD > $OUT/pmc_sq.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc_f64 -- $CMD > $OUT/pmc_f64.log 2>&1 || exit 1
cd $R && timeout -k 10 600 python bench.py > $OUT/bench_default.log 2>&1
tail -1 $OUT/bench_default.log | cut -c1-400
