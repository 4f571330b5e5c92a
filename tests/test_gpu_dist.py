"""N > 1 bench path rehearsed on ONE GPU: two ranks share cuda:0 and talk over gloo (RCCL needs one GPU per rank).
Exercises the code the driver runs at N = 2/4/8 -- shard generation, the pipelined mergers (position-sliced =
default, and the all-reduce form), the device-side slice finalize / germ-max fold -- and checks the merged table
against a single-pass reduction of all shards.  (Over gloo the sliced merger gets its reduce-scatter / all-to-all
from all-reduce / all-gather; the RCCL calls themselves only run on a multi-GPU node.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port() -> str:
    """a TCP port nobody on this host listens on right now (a fixed one collides with a run left over from another test session)"""
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return str(s.getsockname()[1])


@pytest.mark.parametrize("world,merge,group", [(2, "sliced", 1), (3, "sliced", 2), (2, "sliced", 4), (2, "allreduce", 1)])
def test_bench_multirank_path_on_one_gpu(world, merge, group):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", free_port(), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "7", "--warmup", "5",
           "--backend", "gloo", "--check", "--config", "c2", "--merge", merge, "--group", str(group)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "check: merged error table == single-pass error table" in r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    d = json.loads(line)
    assert d["n_gpus"] == world and d["scaling"] == "weak" and d["value"] > 0 and "cpu_baseline" not in d
    assert d["config"]["merge"] == merge


@pytest.mark.parametrize("world", [2, 3, 4])
def test_bench_strong_scaling_line_reports_the_tumour_shard(world):
    """The N > 1 line of a strong-scaling job (a small one: c4s; gloo ranks sharing the GPU) carries what north_star's scaling target is
    worded on -- the tumour shard's own R_VC against all tumours on one GPU (no exchange in it) -- beside the whole-step efficiency and
    the single-cohort (unpipelined) one."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", free_port(), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--steps", "8", "--warmup", "3",
           "--backend", "gloo", "--check", "--config", "c4s"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == world and d["scaling"] == "strong" and d["efficiency"] > 0 and d["strong_base"]["value"] > 0
    ts = d["tumour_shard"]
    assert ts["tumours_per_gpu"] in (64 // world, 64 // world + 1) and ts["poisson_call_ms_slowest_rank"] > 0 and ts["one_gpu_poisson_call_ms"] > 0
    assert abs(ts["efficiency"] - ts["one_gpu_poisson_call_ms"] / ts["poisson_call_ms_slowest_rank"] / world) < 1e-9 and d["tumour_shard_efficiency"] == ts["efficiency"]
    assert d["single_cohort_efficiency"] > 0 and d["communication"]["single_batch_latency_ms"] > 0


def test_rccl_calls_of_the_sliced_merge_on_a_size_one_communicator():
    """The native path of dist.SlicedMerger (reduce_scatter_tensor f64, all_to_all_single f32, all_gather_into_tensor u8)
    on a real RCCL communicator -- of size 1, all a one-GPU box offers -- in the bench's pipeline shape, with
    poisson_call reading the gathered blocks."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", free_port(), os.path.join(ROOT, "tools", "nccl_selftest.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "nccl selftest ok" in r.stdout


def test_bench_line_at_one_gpu_has_the_contract_fields():
    """`python bench.py` (N = 1) on the small configuration: ONE JSON line with the contract's fields, the roofline of the dominant
    kernel, and the self-checks of the blocks behind the timed region (layouts, the one-stream pass against the ranges') all true."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--config", "c2", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-e2e",
           "--sustained", "40", "--cold-batches", "2", "--whole-rounds", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "kernels"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["scaling"] == "none" and d["data"].startswith("synthetic") and "workload" in d["config"] and "model" not in d["config"]
    assert d["value"] > 0 and abs(d["value"] - (d["config"]["positions"] * (d["config"]["normals_total"] + d["config"]["tumours_total"])) / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
    rf = d["roofline"]
    assert rf["bound"] in ("hbm", "mfma") and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    assert "traffic" in rf and 0 < rf["frac"] < 1
    assert all(x["same_outputs"] for x in d["other_record_layouts"])
    assert d["roofline_whole_rounds"]["frac"] > 0
    # config 2 is 157 tiles: two ranges of >= 2 tiles apply; the timed region ran them inside the library, the block behind it repeats the
    # pass on one stream and compares the outputs
    if "position_ranges_fallback" in d["config"]:  # a box on which two streams of the library could not be made to overlap: whole launches, said so
        assert d["config"]["position_ranges"] == 1 and d["config"]["position_ranges_fallback"]["requested"] == 2
    else:
        assert d["config"]["position_ranges"] == 2 and d["ranges"]["n"] == 2 and len(d["ranges"]["error_reduce_ms"]) == 2
        assert d["one_stream"]["same_outputs_as_the_timed_region"] is True and d["roofline_overlapped"]["launches_per_step"] == 2
        assert abs(d["roofline"]["avg_ms"] - max(d["one_stream"]["error_reduce_ms"], d["one_stream"]["poisson_call_ms"])) < 1e-9
    assert 0 < d["roofline_pass"]["frac"] < 1
    assert d["sustained"]["passes"] == 40 and d["cold_hbm"]["batches"] == 2
