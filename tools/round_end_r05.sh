#!/bin/bash
# ON THE GPU BOX: the round's records (profiles/r05): GPU test log, default bench line, rehearsals of the N > 1 path (gloo ranks sharing
# the GPU; the whole path on one rank over real RCCL), config 5 at N = 1, the all-scores A/B, command-line phases.
# tools/collect_profiles.sh makes the rocprofv3 part; `python bench.py --e2e c4` the config-4-sized end-to-end leg.
cd "$(dirname "$0")/.."
O=gpurun_out/r05_end
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.json; echo
python bench.py --ranges 1 --no-e2e --no-cpu-baseline > $O/bench_one_stream.json 2> $O/bench_one_stream.err; tail -c 120 $O/bench_one_stream.json; echo
python bench.py --gpus 2 --backend gloo --check --steps 8 --warmup 2 --config c4s > $O/bench_gloo2_c4s.json 2> $O/bench_gloo2_c4s.err; tail -c 120 $O/bench_gloo2_c4s.json; echo
python bench.py --force-dist --config c4 --steps 16 --warmup 4 --check > $O/bench_force_dist_c4.json 2> $O/bench_force_dist_c4.err; tail -c 120 $O/bench_force_dist_c4.json; echo
python bench.py --config c5 --no-e2e --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 120 $O/bench_c5.json; echo
python tools/full_mode_bench.py > $O/full_mode.log 2>&1; tail -1 $O/full_mode.log
python tools/cli_phases.py --reps 5 > $O/cli_phases.log 2>&1; tail -2 $O/cli_phases.log | cut -c1-300
