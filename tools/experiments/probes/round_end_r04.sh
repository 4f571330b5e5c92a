#!/bin/bash
# ON THE GPU BOX: the round's records (profiles/r04): GPU test log, default bench line, rehearsals of the N > 1 path, config 5
# at N = 1, command-line phases with and without the record cache.  tools/collect_profiles.sh makes the rocprofv3 part.
cd "$(dirname "$0")/../../.."
O=gpurun_out/r04_end
mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -2 $O/pytest_gpu.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.json; echo
python bench.py --gpus 2 --backend gloo --check --steps 8 --warmup 2 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; tail -c 120 $O/bench_gloo2.json; echo
python bench.py --force-dist --config c4 --steps 16 --warmup 4 --check > $O/bench_force_dist_c4.json 2> $O/bench_force_dist_c4.err; tail -c 120 $O/bench_force_dist_c4.json; echo
python bench.py --force-dist --config c4 --steps 16 --warmup 4 --check --wide-sums > $O/bench_force_dist_c4_wide.json 2> $O/bench_force_dist_c4_wide.err; tail -c 120 $O/bench_force_dist_c4_wide.json; echo
python bench.py --config c5 --no-e2e --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 120 $O/bench_c5.json; echo
python tools/cli_phases.py --reps 5 > $O/cli_phases.log 2>&1
python tools/cli_phases.py --reps 4 --configs c3 --env AMPLISOLVE_CACHE=1 > $O/cli_phases_cache.log 2>&1
tail -2 $O/cli_phases_cache.log | cut -c1-300
