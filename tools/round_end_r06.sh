#!/bin/bash
# ON THE GPU BOX: the round's records (profiles/r06): default bench line (with tumour_shard_projection and the reference's callVariants in
# cpu_baseline), the driver's command, the one-stream region, rehearsals of the N > 1 path (gloo ranks sharing the GPU; the whole path on
# one rank over real RCCL), config 5 at N = 1, both command lines beside the reference's own code on fresh panels (error estimation
# incl. cohorts outside the exactness envelope; variant calling beside the reference's callVariants), the kernels against the oracle.
# tools/collect_profiles.sh makes the rocprofv3 part; `python -m pytest tests -m gpu` the test log.  Pieces: $1 = bench | fuzz | all
cd "$(dirname "$0")/.."
O=gpurun_out/r06_end
mkdir -p $O
W=${1:-all}
if [ "$W" = bench ] || [ "$W" = all ]; then
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 200 $O/bench_default.json; echo
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 120 $O/bench_driver_cmd.json; echo
python bench.py --ranges 1 --no-e2e --no-cpu-baseline > $O/bench_one_stream.json 2> $O/bench_one_stream.err; tail -c 120 $O/bench_one_stream.json; echo
python bench.py --gpus 2 --backend gloo --check --steps 8 --warmup 2 --config c4s > $O/bench_gloo2_c4s.json 2> $O/bench_gloo2_c4s.err; tail -c 120 $O/bench_gloo2_c4s.json; echo
python bench.py --force-dist --config c4 --steps 16 --warmup 4 --check > $O/bench_force_dist_c4.json 2> $O/bench_force_dist_c4.err; tail -c 120 $O/bench_force_dist_c4.json; echo
python bench.py --config c5 --no-e2e --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_c5.json 2> $O/bench_c5.err; tail -c 120 $O/bench_c5.json; echo
python bench.py --e2e c4 > $O/e2e_c4.json 2> $O/e2e_c4.err; tail -c 120 $O/e2e_c4.json; echo
python tools/full_mode_bench.py > $O/full_mode.log 2>&1; tail -1 $O/full_mode.log
fi
if [ "$W" = fuzz ] || [ "$W" = all ]; then
python tools/fuzz_cli_vc_vs_reference.py 150 6000 > $O/fuzz_cli_vc_vs_reference.log 2>&1; tail -1 $O/fuzz_cli_vc_vs_reference.log
python tools/fuzz_cli_vs_reference.py 150 31337 > $O/fuzz_cli_vs_reference_31337.log 2>&1; tail -1 $O/fuzz_cli_vs_reference_31337.log
python tools/fuzz_cli_vs_reference.py 120 977 > $O/fuzz_cli_vs_reference_977.log 2>&1; tail -1 $O/fuzz_cli_vs_reference_977.log
python tools/fuzz_parity.py --seconds 240 --seed 83 > $O/fuzz_parity.log 2>&1; tail -1 $O/fuzz_parity.log
fi
