#!/usr/bin/env python3
"""bench.py -- AmpliSolve hot path on MI355X: error estimation + Poisson calling.

One "step" = one pass of the hot path over one synthetic batch that is already
resident in HBM: error_reduce over the rank's normal-sample shard -> (N>1: RCCL
merge of the accumulator table) -> error_finalize -> poisson_call over the
rank's tumour shard.  Workload at N=1: BASELINE.json configs[2], the one the
metric's target is quoted on (100k positions x 256 normals x 96 tumours).
N>1: strong scaling of configs[3] (1024 normals + 1024 tumours split N ways);
started without a launcher, `--gpus N` starts its own N ranks (torch.distributed.run)
before it touches the GPU, and every N>1 line carries the same job's one-GPU time
(strong_base), the efficiency against it and a communication block.

Prints ONE JSON line on rank 0 (contract in the task prompt).  `value` counts
position-evaluations: one (position, sample) record pushed through its half of
the path (normals through the gated reduction, tumours through the Poisson
test), whole job, per second.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6  # vector FP64: 256 CUs x 4 SIMDs x 16 lanes x 2 flops (FMA) x 2.4 GHz = half of the guide's 157.3 TF FP32 vector figure
# (MI355X_MICROARCH.md counts packed FP32, two per lane); v_fma_f64 measured at full issue rate, 4 cycles per wave instruction (tools/micro/op_rates.hip)
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
SEED = 0xA3F15017 + 2   # seed base + config index

CONFIGS = {
    "c2": dict(P=10_000, S=32, T=8, depth=2000, strong=False, name="synthetic 10k positions x 32 normals x 8 tumours"),
    "c3": dict(P=100_000, S=256, T=96, depth=2000, strong=False, name="synthetic 100k positions x 256 normals x 96 tumours (ctDNA-scale)"),
    # BASELINE configs[3]: the WHOLE job is 1024 normals + 1024 tumours; N ranks take 1024/N of each (strong scaling)
    "c4": dict(P=100_000, S=1024, T=1024, depth=2000, strong=True, name="synthetic 100k positions x 1024 normals x 1024 tumours, split over the GPUs"),
    # BASELINE configs[4]: 1 M positions x 256 normals at 50 000x (T is not given there: 64 = 8 per GPU at N = 8); the
    # depth does not fit uint16, so --records auto holds it in 24-byte records; the fixed job is split over the GPUs (strong)
    # a strong-scaling job small enough for tests of the N > 1 line (not a BASELINE configuration)
    "c4s": dict(P=20_000, S=64, T=64, depth=2000, strong=True, name="test job: 20k positions x 64 normals x 64 tumours, split over the GPUs"),
    "c5": dict(P=1_000_000, S=256, T=64, depth=50_000, strong=True, name="synthetic 1M positions x 256 normals x 64 tumours, depth 50000x (VAF 1% stress), split over the GPUs"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="default: c3 on one GPU (the configuration the metric's target is quoted on); with --gpus N > 1: c4, "
                         "BASELINE's 8-GPU job of 1024 normals + 1024 tumours split N ways (strong scaling).  c2 / c3 with N > 1: "
                         "every rank owns a shard of that size (weak scaling)")
    ap.add_argument("--wide-sums", action="store_true", help="N > 1, sliced merge: exchange the sums as 21 plain planes (168 B per position) instead of the 14 packed ones (112 B)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the end-to-end leg (ASEQ text on disk -> tables / calls through the command lines)")
    ap.add_argument("--sustained", type=int, default=2000, help="passes of the sustained-rate block after the timed region (0 = skip)")
    ap.add_argument("--cold-batches", type=int, default=3, help="N = 1: distinct resident batches the cold-HBM block rotates over after the timed region "
                    "(no pass finds its inputs in the Infinity Cache); 0 or 1 = skip")
    ap.add_argument("--ranges", type=int, default=2, help="N = 1: position ranges on concurrent streams INSIDE the library (ampli_set_ranges): "
                    "error_estimate and poisson_call cut the panel into this many tile-aligned ranges, each on a stream of its own, so that "
                    "back-to-back passes fill each other's partly filled rounds of workgroups.  1 = every launch whole, on one stream")
    ap.add_argument("--no-split-ranges", dest="split_ranges", action="store_false", help="the same as --ranges 1: the timed region on one stream, "
                    "per-kernel HIP events around undisturbed launches (what rounds 1-4 timed)")
    ap.add_argument("--whole-rounds", type=int, default=4, help="N = 1: after the timed region, time the dominant reduce kernel on a panel of this many "
                    "whole rounds of resident workgroups (what the partly filled last round of the configuration costs); 0 = skip")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="time the reference's error estimation on ALL normals of config 3 (~6 min on one core)")
    ap.add_argument("--e2e", default=None, choices=list(CONFIGS), help="ONLY the end-to-end leg of that configuration, nothing else: its files written as ASEQ text, "
                    "both executables with their phase clocks, the reference's error estimation on a 6-file sample.  Opt-in because of its size "
                    "(c4: 2048 files, ~13 GB of text); prints one JSON object {\"e2e\": {...}}")
    ap.add_argument("--mode", default="prefilter", choices=["prefilter", "full"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--splits", type=int, default=0, help="error_reduce sample splits (0 = auto)")
    ap.add_argument("--groups", type=int, default=0, help="error_reduce lane groups per wave (0 = auto, 1, 2, 4)")
    ap.add_argument("--event-every", type=int, default=5, help="record the per-kernel HIP events on every n-th timed step (each record costs a few us of stream time)")
    ap.add_argument("--streams", type=int, default=1, help="N = 1 only: independent batches round-robin over this many HIP streams "
                    "(cross-batch overlap; per-kernel times then include contention, so the default stays 1)")
    ap.add_argument("--no-shard-projection", action="store_true", help="N = 1: skip the projection of config 4's tumour shard (poisson_call over 1024 / N tumours, N = 1, 2, 4, 8)")
    ap.add_argument("--async-drain", action="store_true", help="poisson_call's drain kernel on a side stream (measured: no gain on config 3)")
    ap.add_argument("--records", default="auto", choices=["auto", "i32", "u24", "u16"],
                    help="record layout resident in HBM (identical results): i32 = 8 x int32 = 32 B per record; u24 = 8 x 24 bits = "
                         "24 B (counts <= 2^24 - 2, i.e. everything the fast kernels accept); u16 = 8 x uint16 = 16 B (counts <= 65534). "
                         "auto (default) = the narrowest the workload's counts fit -- what the command lines' host packer uploads for the "
                         "same cohort (csrc/host/aseq.cpp).  The cohort is packed once at setup (ampli_records_pack24/16); at N = 1 the "
                         "other layouts are timed too, outside the timed region")
    ap.add_argument("--merge", default="sliced", choices=["sliced", "allreduce"],
                    help="N>1 exchange: sliced = reduce-scatter + all-to-all + all-gather by position slices (default); "
                         "allreduce = one packed all-reduce + all-gather of whole germ-max regions")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--group", type=int, default=0, help="N>1, sliced merge: independent batches per round of collectives "
                    "(fewer, larger RCCL messages and fewer cross-stream waits per batch); 0 = auto: 4 for runs of >= 16 steps, "
                    "2 for >= 8, else 1 (the last group's exchange has nothing to hide behind, so short runs want small groups)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: run the N>1 code path (process group, merge, pipelined loop) "
                    "even with one rank -- over RCCL this exercises the real collectives on a one-GPU box")
    ap.add_argument("--check", action="store_true", help="N>1: verify the merged table against a single-pass reduction of all shards")
    ap.add_argument("--no-strong-base", action="store_true", help="N>1, strong scaling: skip the one-GPU timing of the whole job on rank 0 "
                    "(strong_base / efficiency in the line)")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------
# Files of the synthetic workload (written by the C++ host library, bit-identical to what ampli_synth_fill puts
# into HBM): shared by the end-to-end leg and by the reference's error-estimation run of the CPU baseline.
# ----------------------------------------------------------------------------------------------------
BIN = os.path.join(ROOT, "amplisolve_amd", "bin")


def write_workload_files(d, P, S, T, depth):
    import ctypes as C

    from amplisolve_amd import host_lib

    H = host_lib()
    t0 = time.perf_counter()
    assert H.ampli_host_synth_write_panel(os.path.join(d, "panel.bed").encode(), os.path.join(d, "refbases.txt").encode(), P, SEED) == 0
    nb = H.ampli_host_synth_write_aseq(os.path.join(d, "N").encode(), b"N", P, S, 0, SEED, depth, 0, 0)
    tb = H.ampli_host_synth_write_aseq(os.path.join(d, "T").encode(), b"T", P, T, 0, SEED, depth, 1, 0)
    assert nb > 0 and tb > 0
    open(os.path.join(d, "dups.txt"), "w").close()
    return dict(normal_text_bytes=int(nb), tumour_text_bytes=int(tb), write_s=time.perf_counter() - t0)


def _run_timed(cmd, cwd, env=None):
    """child process with its wall time, the TIMING lines of its stderr and ITS OWN peak RSS (os.wait4)"""
    e = dict(os.environ, **(env or {}))
    with tempfile.TemporaryFile("w+") as fo, tempfile.TemporaryFile("w+") as fe:
        t0, e0 = time.perf_counter(), time.time()
        pr = subprocess.Popen(cmd, cwd=cwd, env=e, stdout=fo, stderr=fe, text=True)
        _, status, ru = os.wait4(pr.pid, 0)
        wall, e1 = time.perf_counter() - t0, time.time()
        pr.returncode = os.waitstatus_to_exitcode(status)
        fo.seek(0), fe.seek(0)
        out, err = fo.read(), fe.read()
    timing = {}
    for ln in err.splitlines():
        if ln.startswith("TIMING2 "):  # the phase clock: name seconds critical|overlapped|total
            w = ln.split()
            timing.setdefault("phases", {})[w[1] + ("*" if w[3] == "overlapped" else "")] = round(float(w[2]), 4)
            continue
        if ln.startswith("TIMING"):
            w = ln.split()
            timing[w[1]] = float(w[2])
            for k, v in zip(w[3::2], w[4::2]):
                try:
                    timing[f"{w[1]}.{k}"] = float(v)
                except ValueError:
                    pass
    ph = timing.get("phases", {})
    if "epoch_at_report" in ph and "wall_in_main" in ph:  # what lies outside main(), split: before it was entered / after its report
        ep = ph.pop("epoch_at_report")
        timing["outside_main"] = {"before_main_s": round(ep - ph["wall_in_main"] - e0, 4), "after_report_s": round(e1 - ep, 4)}
    return pr.returncode, wall, timing, out, err, ru.ru_maxrss / 1024.0


def reference_ee_run(d, normals_dir, out_name):
    """the reference's own error-estimation code (oracle/_ref/ee_ref_driver, compiled -O2 from /root/reference where it
    lies; one core) on a directory of the workload's files.  Short relative paths: the reference has 50-char buffers."""
    from oracle import pyoracle as orc

    drv = orc.REF_EE_DRIVER
    if not os.path.exists(drv):
        return None
    os.makedirs(os.path.join(d, out_name), exist_ok=True)
    rc, wall, tm, out, err, _ = _run_timed([drv, "panel.bed", "refbases.txt", "dups.txt", normals_dir, "0.002", "100", out_name], d)
    if rc != 0 or "storeGermlineStatistics" not in tm:
        return None
    t = tm["storeGermlineStatistics"] + tm["estimateThresholds"] + tm["generateFinalOutput"]
    return dict(records=int(tm["records"]), seconds=t, wall_s=wall, records_per_s=tm["records"] / t, cores=1,
                phases={k: tm[k] for k in ("storeGermlineStatistics", "estimateThresholds", "generateFinalOutput")},
                table=os.path.join(d, out_name, "positionSpecificNoise_0.0020.txt"))


def reference_vc_call_run(d, H, P, depth, table, T_total):
    """the reference's own callVariants (oracle/_ref/vc_call_ref_driver: AmpliSolveVariantCalling.cpp compiled -O2 where it lies minus its Boost
    include, fisherTest and the 12 `p=fisherTest(...)` statements; p stays -1) on tumour files of the workload: `time` mode = storeInputFile ->
    storeCountList -> callVariants as main() runs them (VC:320-344), a clock around each.  One process = one core; then n processes over a
    directory split (tumour files are independent), all started together, wall clock of the slowest."""
    from oracle import pyoracle as orc

    n1 = 6
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    nproc = max(1, min(cores, 16))
    per = 2  # files per process in the split
    need = max(n1, nproc * per)
    if H.ampli_host_synth_write_aseq(os.path.join(d, "T").encode(), b"T", P, need, 0, SEED, depth, 1, 0) <= 0:
        return None
    files = sorted(os.listdir(os.path.join(d, "T")))

    def stage(name, fs):
        os.makedirs(os.path.join(d, name), exist_ok=True)
        for f in fs:
            os.symlink(os.path.join("..", "T", f), os.path.join(d, name, f))
        os.makedirs(os.path.join(d, name + "_o", "AmpliSolveVariantCalling_interm_files"), exist_ok=True)
        return [orc.REF_VC_CALL_DRIVER, "time", table, name, name + "_o", "100", "0.05"]

    rc, wall, tm, out, err, _ = _run_timed(stage("T1", files[:n1]), d)
    if rc != 0 or "callVariants" not in tm:
        return None
    lines1 = sum(sum(1 for _ in open(os.path.join(d, "T", f))) - 1 for f in files[:n1])
    calls1 = sum(1 for _ in open(os.path.join(d, "T1_o", "Summary_Variant_Info.txt"))) - 1
    res = dict(files_1_core=n1, records_1_core=lines1, callVariants_s_1_core=tm["callVariants"], storeInputFile_s=tm["storeInputFile"],
               storeCountList_s=tm.get("storeCountList"), calls_1_core=calls1, tumour_records_per_s_1_core=lines1 / tm["callVariants"], cores=1,
               what="reference callVariants (VC:633-3304) without its Fisher statements, -O2, in main()'s call order (oracle/ref_vc_call_driver.cpp)")
    cmds = [stage(f"S{k:02d}", files[k * per:(k + 1) * per]) for k in range(nproc)]
    t0 = time.perf_counter()
    prs = [subprocess.Popen(c, cwd=d, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True) for c in cmds]
    errs = [pr.communicate()[1] for pr in prs]
    wall_n = time.perf_counter() - t0
    if any(pr.returncode != 0 for pr in prs):
        return res
    cv = [float(ln.split()[2]) for e in errs for ln in e.splitlines() if ln.startswith("TIMING callVariants")]
    lines_n = sum(sum(1 for _ in open(os.path.join(d, "T", f))) - 1 for f in files[:nproc * per])
    res.update(processes=nproc, host_cores=cores, files_n_processes=nproc * per, records_n_processes=lines_n, wall_s_n_processes=wall_n,
               callVariants_s_slowest_process=max(cv) if cv else None,
               tumour_records_per_s_n_processes=lines_n / max(cv) if cv else lines_n / wall_n,
               note_n="callVariants' own seconds in the slowest of the processes (each also loads the table: storeInputFile_s, once per process, inside wall_s_n_processes)")
    return res


def e2e_leg(name, cfg, ref_files=None, keep_dir=None):
    """ASEQ text on disk -> positionSpecificNoise table -> Summary / VCFs through the two command lines, timed; the
    reference's error estimation on the same files beside it (all of them when ref_files is None)."""
    P, S, T, depth = cfg["P"], cfg["S"], cfg["T"], cfg["depth"]
    d = keep_dir or tempfile.mkdtemp(prefix=f"ampli_e2e_{name}_")
    try:
        w = write_workload_files(d, P, S, T, depth)
        env = {"AMPLISOLVE_TIMING": "1", "AMPLISOLVE_STRICT_EXIT": "1", "AMPLISOLVE_REFBASES_FILE": "refbases.txt"}
        rc, ee_wall, ee_t, out, err, ee_rss = _run_timed([os.path.join(BIN, "AmpliSolveErrorEstimation"), "panel_design=panel.bed", "reference_genome=unused.fa",
                                                            "germline_dir=N", "C_value=0.002", "coverage_cutoff=100", "default_error=0.01", "output_dir=ee"], d, env)
        if rc != 0:
            return {"error": "AmpliSolveErrorEstimation failed: " + (out + err)[-400:]}
        table = os.path.join(d, "ee", "positionSpecificNoise_0.0020.txt")
        rc, vc_wall, vc_t, out, err, vc_rss = _run_timed([os.path.join(BIN, "AmpliSolveVariantCalling"), "errorFile=ee/positionSpecificNoise_0.0020.txt",
                                                            "tumour_dir=T", "output_dir=vc", "coverage_cutoff=100", "p_value=0.05"], d, env)
        if rc != 0:
            return {"error": "AmpliSolveVariantCalling failed: " + (out + err)[-400:]}
        n_rec, t_rec = int(ee_t.get("stream.lines", 0)), int(vc_t.get("stream.lines", 0))
        res = {
            "workload": cfg["name"], "text_bytes": w["normal_text_bytes"] + w["tumour_text_bytes"], "files": S + T,
            "error_estimation": {"wall_s": ee_wall, "records": n_rec, "records_per_s": n_rec / ee_wall, "text_GB_per_s": w["normal_text_bytes"] / ee_wall / 1e9,
                                 "record_array_MB": ee_t.get("stream.record_MB"), "bytes_per_record_uploaded": (ee_t.get("stream.record_MB", 0) * 1e6 / max(1, P * S)),
                                 "host_peak_rss_MB": ee_rss,
                                 "phases_s": {"panel": ee_t.get("panel"), "stream(parse+upload+reduce)": ee_t.get("stream"), "parser_busy": ee_t.get("stream.parse_busy"),
                                              "waiting_for_gpu": ee_t.get("stream.device_wait"), "table_write": ee_t.get("write")},
                                 "chunks": int(ee_t.get("stream.chunks", 0)),
                                 # which error_reduce kernel each chunk's launch was (ampli_last_reduce_kernel, printed by the executable)
                                 "reduce_launches": {"error_reduce_u16_kernel": int(ee_t.get("reduce_launches.error_reduce_u16_kernel", 0)),
                                                     "error_reduce_u24_kernel": int(ee_t.get("reduce_launches.error_reduce_u24_kernel", 0)),
                                                     "error_reduce_kernel": int(ee_t.get("reduce_launches.error_reduce_kernel", 0)),
                                                     "accumulator_table": bool(ee_t.get("reduce_launches.accumulator_table", 0))},
                                 # where the wall time goes: the phase clock of the executable (csrc/host/pipeline.cpp PhaseClock).
                                 # Entries without * lie on the main thread's path and add up to wall_in_main; entries with * ran
                                 # on other threads beside it (runtime start-up, parsers, by-product files, ring teardown)
                                 "breakdown_s": ee_t.get("phases"),
                                 "outside_main_s": round(ee_wall - ee_t.get("phases", {}).get("wall_in_main", ee_wall), 4), "outside_main": ee_t.get("outside_main")},
            "variant_calling": {"wall_s": vc_wall, "records": t_rec, "records_per_s": t_rec / vc_wall, "calls": int(vc_t.get("stream.calls", 0)),
                                "host_peak_rss_MB": vc_rss,
                                "phases_s": {"table_read": vc_t.get("table"), "stream(parse+upload+call)": vc_t.get("stream"), "parser_busy": vc_t.get("stream.parse_busy"),
                                             "annotate+write": vc_t.get("annotate+write")},
                                "breakdown_s": vc_t.get("phases"),
                                "outside_main_s": round(vc_wall - vc_t.get("phases", {}).get("wall_in_main", vc_wall), 4), "outside_main": vc_t.get("outside_main")},
            "records_per_s": (n_rec + t_rec) / (ee_wall + vc_wall),
            "note": "wall clock of the two executables incl. process start and HIP runtime start-up (0.05-0.25 s each: breakdown_s.runtime_init*; "
                    "the parsers run beside it); the host packer uploads the narrowest record layout the counts fit (bytes_per_record_uploaded)",
        }
        # the reference's error estimation on the same files (a subset directory of symlinks when the cohort is large)
        ref_dir = "N"
        if ref_files is not None and ref_files < S:
            ref_dir = "Nref"
            os.makedirs(os.path.join(d, ref_dir), exist_ok=True)
            for f in sorted(os.listdir(os.path.join(d, "N")))[:ref_files]:
                os.symlink(os.path.join("..", "N", f), os.path.join(d, ref_dir, f))
        ref = reference_ee_run(d, ref_dir, "ref")
        if ref:
            # same files, same directory literal -> same visit order: the tables must be byte-identical.  For a subset of the
            # cohort our executable runs once more, on that subset directory, so that there is a table to compare with
            ours = table
            if ref_dir != "N":
                rc, _, _, out, err, _ = _run_timed([os.path.join(BIN, "AmpliSolveErrorEstimation"), "panel_design=panel.bed", "reference_genome=unused.fa",
                                                    f"germline_dir={ref_dir}", "C_value=0.002", "coverage_cutoff=100", "default_error=0.01", "output_dir=ee_sub"], d, env)
                ours = os.path.join(d, "ee_sub", "positionSpecificNoise_0.0020.txt") if rc == 0 else None
            same = (open(ref["table"], "rb").read() == open(ours, "rb").read()) if ours and os.path.exists(ours) else None
            res["reference_error_estimation"] = {
                "records": ref["records"], "seconds": ref["seconds"], "records_per_s": ref["records_per_s"], "cores": 1, "phases_s": ref["phases"],
                "files": ref_files if ref_dir != "N" else S, "table_identical_to_ours": same,
                "what": "reference AmpliSolveErrorEstimation.cpp compiled -O2 where it lies (oracle/_ref/ee_ref_driver: storeGermlineStatistics + "
                        "estimateThresholds + generateFinalOutput; its per-position samtools step is not on the path and is left out)",
                "extrapolated_full_cohort_s": ref["seconds"] * (n_rec / ref["records"]) if ref_dir != "N" else None,
                "extrapolation": "linear in the number of records; the one full-size run of the reference (profiles/r02/bench_cpu_baseline_full.json) took "
                                 "112.5 s, so the rule errs on the reference's side by 15-30 %" if ref_dir != "N" else None,
            }
            res["error_estimation"]["speedup_vs_reference_records_per_s"] = res["error_estimation"]["records_per_s"] / ref["records_per_s"]
        return res
    finally:
        if keep_dir is None:
            import shutil

            shutil.rmtree(d, ignore_errors=True)


# ----------------------------------------------------------------------------------------------------
# CPU baseline leg (rank 0, N = 1): the reference's own code (error estimation: the whole translation unit; calling: the
# whole translation unit but its Boost include, fisherTest and the 12 statements that call it -- oracle/Makefile), on a
# bounded sample of the SAME workload, plus the oracle port on both halves.
# ----------------------------------------------------------------------------------------------------
def cpu_baseline(cfg, thr, ref_code, full=False):
    import numpy as np

    from oracle import pyoracle as orc
    from tests.helpers import synth_recs, synth_ref

    P, S, T, depth = cfg["P"], cfg["S"], cfg["T"], cfg["depth"]
    out = {"unit": "position-evaluations/s"}
    # --- port (oracle) on both halves, one core ---------------------------------------------------
    Pp, Sp, Tp = 50_000, 64, 24
    normals = synth_recs(Pp, Sp, seed=SEED, depth=depth)
    tumours = synth_recs(Pp, Tp, seed=SEED, depth=depth, tumour=True)
    t0 = time.perf_counter()
    fin = orc.error_finalize(orc.error_reduce(normals, Pp, 0.002, 100))
    t1 = time.perf_counter()
    orc.poisson_call(tumours, Pp, fin["thr"], synth_ref(Pp, seed=SEED), 100, dense=False)
    t2 = time.perf_counter()
    port = dict(value=(Pp * Sp + Pp * Tp) / (t2 - t0), unit="position-evaluations/s", cores=1, kind="port",
                sample=f"oracle/ampli_oracle.c on {Pp} positions x {Sp} normals + {Tp} tumours of the same synthetic panel "
                       f"(error-est {t1 - t0:.3f} s, calling {t2 - t1:.3f} s)")
    del normals, tumours
    # --- the reference's error estimation: a bounded sample of the workload's own normal files ------------
    ee = None
    n_ref = S if full else 12
    d = tempfile.mkdtemp(prefix="ampli_cpu_")
    try:
        import ctypes as C

        from amplisolve_amd import host_lib

        H = host_lib()
        H.ampli_host_synth_write_panel(os.path.join(d, "panel.bed").encode(), os.path.join(d, "refbases.txt").encode(), P, SEED)
        H.ampli_host_synth_write_aseq(os.path.join(d, "N").encode(), b"N", P, n_ref, 0, SEED, depth, 0, 0)
        open(os.path.join(d, "dups.txt"), "w").close()
        r = reference_ee_run(d, "N", "ref")
        vc_load = None
        vc_call = None
        if r and os.path.exists(getattr(orc, "REF_VC_DRIVER", "")):
            # the calling half's table load, by the reference's own storeInputFile (VC:430-576, compiled in place without Boost) on the
            # table the reference just wrote for this panel: P rows, four threshold and four germ-max cells each
            try:
                rc_, _, tm_, _, _, _ = _run_timed([orc.REF_VC_DRIVER, "time", os.path.relpath(r["table"], d), "dummy.vcf"], d)
                if rc_ == 0 and "storeInputFile" in tm_:
                    vc_load = dict(seconds=tm_["storeInputFile"], positions=int(tm_.get("storeInputFile.positions", 0)), cores=1,
                                   what="reference storeInputFile (VC:430-576) on the workload's error table, once per AmpliSolveVariantCalling run")
            except Exception:  # noqa: BLE001
                vc_load = None
        if r and os.path.exists(getattr(orc, "REF_VC_CALL_DRIVER", "")):
            vc_call = reference_vc_call_run(d, H, P, depth, os.path.relpath(r["table"], d), T)
        if r:
            ee = dict(records=r["records"], seconds=r["seconds"], records_per_s=r["records_per_s"], cores=1, phases_s=r["phases"],
                      sample=f"{n_ref} of the {S} normal files of the workload ({P} positions each), reference AmpliSolveErrorEstimation.cpp -O2, "
                             "samtools step left out",
                      extrapolated_whole_cohort_s=r["seconds"] * S / n_ref,
                      extrapolation="seconds x (normals / files timed): linear in records; the one full-size run (--cpu-baseline-full, "
                                    "profiles/r02/bench_cpu_baseline_full.json) took 112.5 s, so the rule errs on the reference's side by ~15 %")
    finally:
        import shutil

        shutil.rmtree(d, ignore_errors=True)
    out["port"] = port
    out["ee"] = ee
    out["vc_callvariants"] = vc_call
    out["vc_table_load"] = vc_load
    if ee and vc_call:
        # one figure in the metric's unit: the workload's record mix through the reference's own code on one core
        t_load = vc_call["storeInputFile_s"]
        t_mix = P * S / ee["records_per_s"] + P * T / vc_call["tumour_records_per_s_1_core"] + t_load
        out.update(value=(P * S + P * T) / t_mix, cores=1, kind="reference",
                   value_all_cores=(P * S + P * T) / (P * S / ee["records_per_s"] + P * T / vc_call["tumour_records_per_s_n_processes"] + t_load),
                   sample=f"reference code on one host core, bounded sample of the same workload: error estimation (storeGermlineStatistics + estimateThresholds + "
                          f"generateFinalOutput) {ee['records']} records in {ee['seconds']:.1f} s; calling = the reference's own callVariants (VC:633-3304, compiled in place "
                          f"without its 12 fisherTest statements: oracle/Makefile VC_CALL_DROP) on {vc_call['files_1_core']} of the {T} tumour files, "
                          f"{vc_call['records_1_core']} records in {vc_call['callVariants_s_1_core']:.1f} s, + its table load (storeInputFile, {t_load:.2f} s once); value = (normal + "
                          f"tumour records of the workload) / (their time at those rates).  value_all_cores: the calling half as {vc_call['processes']} processes over a "
                          f"directory split (tumour files are independent; the error estimation cannot use more than one core)")
    else:
        out.update(value=port["value"], cores=1, kind="port", sample=port["sample"])
    return out


def kernel_source_sha():
    import hashlib

    h = hashlib.sha256()
    for f in ("ampli_kernels.hip", "ampli_math.h"):
        h.update(open(os.path.join(ROOT, "amplisolve_amd", "csrc", f), "rb").read())
    return h.hexdigest()


# ----------------------------------------------------------------------------------------------------
def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as fresh child processes (one per
    GPU, torch.distributed.run, rendezvous on 127.0.0.1) BEFORE this process makes any GPU call, pass rank 0's JSON line
    through, and end with the children's status.  Never falls through to one rank."""
    import socket

    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print(f"bench.py: --gpus {args.gpus} without a launcher: starting {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    pr = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in pr.stdout:
        if ln.lstrip().startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = pr.wait()
    if rc != 0:
        raise SystemExit(f"bench.py: the {args.gpus}-rank job ended with status {rc}")
    if line is None:
        raise SystemExit(f"bench.py: the {args.gpus}-rank job printed no result line")
    d = json.loads(line)
    if d.get("n_gpus") != args.gpus:
        raise SystemExit(f"bench.py: asked for {args.gpus} ranks, the job reports n_gpus = {d.get('n_gpus')}")
    print(line, flush=True)


def main():
    args = parse_args()
    if args.e2e:
        cfg = CONFIGS[args.e2e]
        t0 = time.perf_counter()
        leg = e2e_leg(args.e2e, cfg, ref_files=6)
        leg["leg_wall_s"] = time.perf_counter() - t0
        ee, vc = leg.get("error_estimation", {}), leg.get("variant_calling", {})
        if ee and vc:  # what bounds a cohort of this size: the largest critical phase of each command line
            for side in (ee, vc):
                crit = {k: v for k, v in (side.get("breakdown_s") or {}).items() if not k.endswith("*") and k not in ("wall_in_main", "unattributed")}
                side["largest_critical_phases"] = sorted(crit.items(), key=lambda kv: -kv[1])[:4]
        print(json.dumps({"e2e": {args.e2e: leg}}))
        return 0
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)  # no GPU call has been made in this process
    # ONE JSON line on stdout and nothing else: RCCL prints a version banner to stdout when its first communicator comes up,
    # and a line in front of the result is a line a parser may take for it.  File descriptor 1 points at stderr from here on
    # (C libraries included); the result goes to the saved descriptor at the very end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.config is None:
        args.config = "c4" if max(world, args.gpus) > 1 else "c3"
    cfg = CONFIGS[args.config]
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the rank count of the line must be the one that ran")

    multi = world > 1 or args.force_dist  # the N>1 code path (also runs with one rank under --force-dist)
    import torch
    import torch.distributed as dist

    from amplisolve_amd import Context
    from amplisolve_amd.api import POISSON_FULL, POISSON_PREFILTER
    from amplisolve_amd.dist import merge_error_table

    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)  # == local_rank on a full node; ranks share a device only in rehearsals
    torch.cuda.set_device(dev_index)
    if multi:
        if world == 1 and "RANK" not in os.environ:  # --force-dist without a launcher: a process group of one
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29811")
            os.environ["RANK"], os.environ["WORLD_SIZE"], os.environ["LOCAL_RANK"] = "0", "1", "0"
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    # everything (kernels, torch plumbing, RCCL's stream dependencies) hangs off ONE non-default stream: the legacy
    # null stream carries implicit synchronisation that costs a few us per launch
    main_stream = torch.cuda.Stream(device=dev_index)
    torch.cuda.set_stream(main_stream)
    ctx = Context(dev_index)
    if args.splits or args.groups:
        ctx.set_tuning(args.splits, groups=args.groups)
    mode = POISSON_PREFILTER if args.mode == "prefilter" else POISSON_FULL

    P, depth = cfg["P"], cfg["depth"]
    from amplisolve_amd.dist import shard_range

    if cfg["strong"]:  # the job is fixed: every rank takes its contiguous share of the samples (visit order)
        s_lo, s_hi = shard_range(cfg["S"], rank, world)
        t_lo, t_hi = shard_range(cfg["T"], rank, world)
        S, T = s_hi - s_lo, t_hi - t_lo
        S_total, T_total = cfg["S"], cfg["T"]
        if S < 1 or T < 1:
            raise SystemExit(f"{args.config}: fewer samples than ranks")
    else:  # every rank owns a shard of the configuration's size
        S, T = cfg["S"], cfg["T"]
        s_lo, t_lo = rank * S, rank * T
        S_total, T_total = S * world, T * world
    # synthetic shard of this rank, generated in HBM (bit-identical to the host generator)
    normals = ctx.synth_fill(P, S, first_sample=s_lo, seed=SEED, depth=depth)
    tumours = ctx.synth_fill(P, T, first_sample=t_lo, seed=SEED, depth=depth, tumour=True)
    ref_code = ctx.synth_ref(P, seed=SEED)
    # record layout: a resident cohort is packed ONCE (data preparation, like the H2D copy it replaces) and then
    # evaluated many times; identical results in every layout (tests/test_gpu_u16.py)
    REC_BYTES = {"i32": 32, "u24": 24, "u16": 16}
    packed = {"i32": (normals, tumours)}
    for name in ("u24", "u16"):
        pn, ok_n = ctx.pack(normals, name)
        pt, ok_t = ctx.pack(tumours, name)
        fits = torch.tensor([1 if (ok_n and ok_t) else 0], dtype=torch.int32, device=ctx.device)
        if multi:
            dist.all_reduce(fits, op=dist.ReduceOp.MIN)  # one layout for the whole job
        if int(fits.item()):
            packed[name] = (pn, pt)
        del pn, pt
    # auto: the narrowest layout the counts fit, which is what the command lines' host packer uploads for this cohort
    layout = args.records if args.records != "auto" else ("u16" if "u16" in packed else "u24" if "u24" in packed else "i32")
    if layout not in packed:
        raise SystemExit(f"--records {layout}: a count of this workload does not fit that layout")
    normals, tumours = packed[layout]
    ctx.set_record_layout(layout)
    if multi or args.streams > 1:
        packed = {layout: packed[layout]}  # the other layouts are only kept for the N = 1 comparison legs
    rec_bytes = REC_BYTES[layout]
    accs = [ctx.new_acc(P) for _ in range(2 if multi else 1)]
    for a in accs:
        a.buf.zero_()
    acc = accs[0]
    fin = None
    call_mask = torch.empty((T, P), dtype=torch.uint8, device=ctx.device)
    cap = 1 << 20
    from amplisolve_amd._lib import Call
    from amplisolve_amd.api import CALL_COUNTER_STRIDE, CALL_COUNTER_WORDS
    from amplisolve_amd.dist import TableMerger
    import ctypes

    calls_buf = torch.empty((cap * ctypes.sizeof(Call),), dtype=torch.uint8, device=ctx.device)
    n_calls = torch.zeros((CALL_COUNTER_WORDS,), dtype=torch.int64, device=ctx.device)
    from amplisolve_amd.dist import SlicedMerger

    sliced = multi and args.merge == "sliced"
    G = args.group if args.group > 0 else (4 if args.steps >= 16 else 2 if args.steps >= 8 else 1)  # batches per round of collectives
    merger = None
    # sums of the sliced exchange: 14 packed planes (112 B per position) unless a shard's values do not fit their share of a
    # packed field -- the kernels flag that (AMPLI_FLAG_SLICE_RANGE) in warm-up and every rank then switches to the 21 plain planes
    slim = sliced and not args.wide_sums
    if multi:
        ctx.set_slice_format(slim)
        merger = (SlicedMerger(P, world, rank, ctx.device, batches=G, slim=slim) if sliced else
                  TableMerger(P, world, ctx.device, ctx.gm_merge, pack=ctx.acc_pack, unpack=ctx.acc_unpack))

    ev = [[ctx.event() for _ in range(4)] for _ in range(args.steps)]
    if not multi:  # nothing happens between the end of the reduce and the start of the call on one GPU: one marker serves as both
        for e in ev:
            e[2] = e[1]
    ev_steps = [i for i in range(args.steps) if i % max(1, args.event_every) == 0]
    # position ranges inside the library (N = 1): the kernels of a pass are then several launches on several streams, timed by events
    # on each range's own stream (ampli_range_event_record: does not close the section)
    n_ranges = args.ranges if (args.split_ranges and not multi and args.streams <= 1 and mode == POISSON_PREFILTER and not args.async_drain) else 1
    if n_ranges > 1 and ((P + 63) // 64 < 2 * n_ranges or P % 4 != 0 or layout == "i32"):
        n_ranges = 1  # the library would run such launches whole (include/amplisolve_hip.h, ampli_set_ranges)
    evr = {i: [[ctx.event() for _ in range(3)] for _ in range(n_ranges)] for i in ev_steps} if n_ranges > 1 else None

    # two error tables, used alternately: with the asynchronous drain the survivors of batch i are still being scored
    # (reading batch i's thresholds) while batch i+1's table is being written
    fins = [None, None]
    if args.async_drain:
        ctx.set_async_drain(True)

    last_blocks = [None]

    def reduce_part(i, timed, slot):
        nonlocal fin
        timed = timed and i % max(1, args.event_every) == 0
        if timed and n_ranges > 1:
            for k in range(n_ranges):
                ctx.range_record(k, evr[i][k][0])
        elif timed:
            ctx.record(ev[i][0])
        if not multi:  # the panel lives on one device: finalize fused into the reduce epilogue (ampli_error_estimate)
            fins[i & 1] = fin = ctx.error_estimate(normals, P, 0.002, 100, out=fins[i & 1])
            if timed and n_ranges > 1:
                for k in range(n_ranges):
                    ctx.range_record(k, evr[i][k][1])
                return
        elif sliced:  # shard of a multi-GPU panel: sums and germ-max pairs straight into the slice-major exchange buffers
            ctx.error_reduce_sliced(normals, P, world, merger.sums[slot], merger.gm[slot], 0.002, 100, first_sample=s_lo)
        else:  # shard of a multi-GPU panel: sums straight into the all-reduce buffer, gm planes into the table
            ctx.error_reduce_packed(normals, P, accs[slot], merger.packed[slot], 0.002, 100, first_sample=s_lo)
        if timed:
            ctx.record(ev[i][1])

    def call_part(i, timed, slot):
        nonlocal fin
        timed = timed and i % max(1, args.event_every) == 0
        if sliced:
            # poisson_call reads the thresholds straight from the gathered blocks (the blocks ARE the error table, by
            # position slice); the plane-major form is only materialised where somebody wants it (--check, the flags)
            if timed:
                ctx.record(ev[i][2])
            ctx.poisson_call(tumours, P, merger.blocks[slot], ref_code, 100, mode=mode, call_mask=call_mask, capacity=cap,
                             calls_buf=calls_buf, n_calls=n_calls, blocks_of=world)
            if timed:
                ctx.record(ev[i][3])
            last_blocks[0] = (merger.blocks[slot], i % G)
            return
        elif multi:  # finalize straight from the all-reduced sums + gathered germ-max regions
            fins[i & 1] = fin = ctx.error_finalize_merged(P, merger.packed[slot], merger.gathered[slot], world, 0.002, 100, out=fins[i & 1])
        if timed and multi:
            ctx.record(ev[i][2])
        ctx.poisson_call(tumours, P, fin.thr, ref_code, 100, mode=mode, call_mask=call_mask, capacity=cap,
                         calls_buf=calls_buf, n_calls=n_calls)
        if timed and n_ranges > 1:
            for k in range(n_ranges):
                ctx.range_record(k, evr[i][k][2])
        elif timed:
            ctx.record(ev[i][3])

    def run_steps(n, timed):
        """n passes of the hot path.  N == 1: strictly sequential.  N > 1: software-pipelined across the independent
        batches -- the table merge of batch i (RCCL, own stream) overlaps finalize + poisson_call of batch i-1 and
        error_reduce of batch i+1; every batch still goes through every stage inside the timed region."""
        if n <= 0:
            return
        if not multi:
            for i in range(n):
                reduce_part(i, timed, 0)
                call_part(i, timed, 0)
            return
        if sliced:
            # three GROUPS of G batches in flight: G x reduce -> [reduce-scatter + all-to-all](group) | G x finalize_slice ->
            # [all-gather](group - 1) | G x poisson_call(group - 2); one round of collectives serves the G batches of a group,
            # and each collective has G whole error_reduce launches between its start and its wait
            hx, hg = {}, {}
            ngroups = (n + G - 1) // G

            def members(gi):
                return range(gi * G, min(n, (gi + 1) * G))

            def mid(gi):
                sj = gi % 3
                merger.wait(hx.pop(gi))
                for i in members(gi):
                    ctx.set_slice_group(G, i % G)
                    ctx.error_finalize_slice(P, world, rank, merger.sum_slice[sj], merger.gm_recv[sj], merger.block[sj], 0.002, 100)
                hg[gi] = merger.start_gather(sj)

            def last(gi):
                merger.wait(hg.pop(gi))
                for i in members(gi):
                    ctx.set_slice_group(G, i % G)
                    call_part(i, timed, gi % 3)

            # issue order on RCCL's (in-order) stream: all-gather(group - 1) BEFORE the exchange of this group, so that the
            # poisson_calls of group - 1 do not wait behind this group's reduce-scatter -- in the steady state and,
            # above all, at the end of the run, where the last exchange then hides behind the previous group's calls
            for gi in range(ngroups):
                for i in members(gi):
                    ctx.set_slice_group(G, i % G)
                    reduce_part(i, timed, gi % 3)
                if gi >= 1:
                    mid(gi - 1)
                hx[gi] = merger.start_exchange(gi % 3)
                if gi >= 2:
                    last(gi - 2)
            if ngroups >= 2:
                last(ngroups - 2)
            mid(ngroups - 1)
            last(ngroups - 1)
            return
        pending = None
        for i in range(n):
            slot = i & 1
            reduce_part(i, timed, slot)
            h = merger.start(accs[slot], slot, prepacked=True)
            if pending is not None:
                j, pslot, ph = pending
                merger.wait(ph)
                call_part(j, timed, pslot)
            pending = (i, slot, h)
        j, pslot, ph = pending
        merger.wait(ph)
        call_part(j, timed, pslot)

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    lanes = None
    if not multi and args.streams > 1:
        # extra lanes: own stream + own outputs each; inputs are shared (read-only)
        lanes = []
        for _ in range(args.streams):
            st = torch.cuda.Stream(device=dev_index)
            with torch.cuda.stream(st):
                c = Context(dev_index)
                c.set_record_layout(layout)
                f = c.error_estimate(normals, P, 0.002, 100)
                r = c.poisson_call(tumours, P, f.thr, ref_code, 100, mode=mode, capacity=cap)
            lanes.append((c, f, r))
        torch.cuda.synchronize()

        def run_steps(n, timed):  # noqa: F811  (replaces the single-stream loop)
            for i in range(n):
                c, f, r = lanes[i % len(lanes)]
                t = timed and i % max(1, args.event_every) == 0
                if t:
                    c.record(ev[i][0])
                c.error_estimate(normals, P, 0.002, 100, out=f)
                if t:
                    c.record(ev[i][1])
                    c.record(ev[i][2])
                c.poisson_call(tumours, P, f.thr, ref_code, 100, mode=mode, call_mask=r["call_mask"], capacity=r["capacity"],
                               calls_buf=r["calls_buf"], n_calls=r["n_calls"])
                if t:
                    c.record(ev[i][3])

    if sliced:
        # The sliced exchange needs reduce-scatter / all-to-all / all-gather on device tensors.  A runtime that lacks one
        # rejects it at the call, on every rank alike: probe them on tiny tensors and AGREE on the outcome (all-reduce of
        # the failure flag) before any rank enters the pipeline -- a fall-back decided by one rank alone would leave the
        # ranks issuing different collectives.  Any other failure later on ends the job.
        bad = torch.zeros(1, dtype=torch.int32, device=ctx.device)
        try:
            merger.probe()
        except Exception as exc:  # noqa: BLE001
            print(f"rank {rank}: sliced exchange unavailable ({type(exc).__name__}: {exc})", file=sys.stderr)
            bad += 1
        dist.all_reduce(bad, op=dist.ReduceOp.MAX)
        if int(bad.item()):
            if rank == 0:
                print("sliced merge rejected by the runtime on some rank: every rank falls back to --merge allreduce", file=sys.stderr)
            sliced = False
            args.merge = "allreduce"
            merger = TableMerger(P, world, ctx.device, ctx.gm_merge, pack=ctx.acc_pack, unpack=ctx.acc_unpack)
    ranges_fallback = None
    if n_ranges > 1:
        ctx.set_ranges(n_ranges)
        if not ctx.ranges_concurrent() or os.environ.get("BENCH_FORCE_RANGES_FALLBACK"):  # (the variable: a rehearsal of this branch)
            # HIP put two of the ranges' streams on one hardware queue and eight replacements did not help (ampli_set_ranges): such
            # ranges run one after the other, slower than whole launches on one stream -- time the one-stream pass instead
            ranges_fallback = {"requested": n_ranges, "used": 1, "reason": "ampli_ranges_concurrent() == 0: the ranges' streams were not seen to overlap on this box"}
            print(f"position ranges: the streams of {n_ranges} ranges do not overlap on this box; the timed region runs whole launches on one stream", file=sys.stderr)
            ctx.set_ranges(1)
            n_ranges = 1
    run_steps(args.warmup, False)
    fence()
    if sliced and slim:
        own = torch.tensor([1 if (ctx.flags(clear=False) & 8) else 0], dtype=torch.int32, device=ctx.device)
        dist.all_reduce(own, op=dist.ReduceOp.MAX)
        if int(own.item()):
            if rank == 0:
                print("slim exchange format: a shard's sums do not fit their share of a packed field -- every rank switches to the wide format", file=sys.stderr)
            ctx.flags(clear=True)
            slim = False
            ctx.set_slice_format(False)
            merger = SlicedMerger(P, world, rank, ctx.device, batches=G, slim=False)
            run_steps(args.warmup, False)
            fence()

    def materialise():
        """sliced merge: the plane-major table of the last finished batch (outside the per-batch work)"""
        nonlocal fin
        if sliced and last_blocks[0] is not None:
            blocks, g = last_blocks[0]
            ctx.set_slice_group(G, g)
            fin = ctx.error_table_unslice(P, world, blocks)
            ctx.set_slice_group(1, 0)
            fins[(args.warmup - 1) & 1] = fin

    materialise()
    if multi and args.check:
        # every shard regenerated locally and reduced in one pass must equal the merged table, bit for bit
        allrecs = ctx.synth_fill(P, S_total, first_sample=0, seed=SEED, depth=depth)
        allrecs, _ = ctx.pack(allrecs, layout)
        ref = ctx.error_estimate(allrecs, P, 0.002, 100)
        got = fins[(args.warmup - 1) & 1]
        present = ref.germ_present > 0
        for name, ok in (("rate", torch.equal(ref.rate, got.rate)), ("code", torch.equal(ref.code, got.code)),
                         ("thr", torch.equal(ref.thr, got.thr)), ("germ_present", torch.equal(ref.germ_present, got.germ_present)),
                         ("germ_val", torch.equal(ref.germ_val[present], got.germ_val[present]))):
            if not ok:
                raise SystemExit(f"rank {rank}: merged error table differs from the single-pass one in {name}")
        del allrecs, ref
        if rank == 0:
            print("check: merged error table == single-pass error table (rate, code, thr, germ-max), bit for bit", file=sys.stderr)
        fence()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        te = torch.tensor([elapsed], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    materialise()
    kflags = ctx.flags(clear=True)
    if kflags != 0:  # AMPLI_FLAG_QUEUE_OVERFLOW / AMPLI_FLAG_RERUN_GENERAL: the passes just timed were not full passes
        raise SystemExit(f"rank {rank}: kernel flags {kflags:#x} were raised inside the timed region; the measurement is void")
    ranges_block = None
    if n_ranges > 1:
        # per range and kernel, under the overlap the ranges exist for: events on the range's own stream.  A launch's duration is then
        # that of a kernel sharing the chip with the other ranges' kernels; their mean is what `rocprofv3 --stats` averages too.
        tiles_all = (P + 63) // 64
        cuts = [min(P, (tiles_all * k // n_ranges) * 64) for k in range(n_ranges)] + [P]
        r_red = [sum(ctx.elapsed_ms(evr[i][k][0], evr[i][k][1]) for i in ev_steps) / len(ev_steps) for k in range(n_ranges)]
        r_call = [sum(ctx.elapsed_ms(evr[i][k][1], evr[i][k][2]) for i in ev_steps) / len(ev_steps) for k in range(n_ranges)]
        t_red, t_call_main = sum(r_red) / n_ranges, sum(r_call) / n_ranges  # mean launch (each launch = 1 / n of the panel)
        ranges_block = {"n": n_ranges, "streams_seen_to_overlap": ctx.ranges_concurrent(), "positions": [cuts[k + 1] - cuts[k] for k in range(n_ranges)], "error_reduce_ms": r_red, "poisson_call_ms": r_call,
                        "note": "ampli_set_ranges: every range's error_estimate + poisson_call on a stream of its own inside the library; durations by "
                                "events on the range's stream (ampli_range_event_record), i.e. of launches that share the chip with the other ranges'"}
        # the outputs the ranges left behind (last pass of the timed region), to be compared with the one-stream pass's below
        ranged_out = (fin.thr.clone(), fin.code.clone(), fin.germ_present.clone(), call_mask.clone(), int(n_calls[::CALL_COUNTER_STRIDE].sum().item()))
        ctx.set_ranges(1)  # everything below times whole launches on one stream
    else:
        t_red = sum(ctx.elapsed_ms(ev[i][0], ev[i][1]) for i in ev_steps) / len(ev_steps)
        t_call_main = sum(ctx.elapsed_ms(ev[i][2], ev[i][3]) for i in ev_steps) / len(ev_steps)  # main-stream part (all of it unless --async-drain)
    ctx.wait_calls()
    if lanes is not None:
        n_calls, fin = lanes[0][2]["n_calls"], lanes[0][1]
    n_found = int(n_calls[::CALL_COUNTER_STRIDE].sum().item())
    # pairs within 1e-6 of the gate Q >= 5 (AMPLI_CALL_BORDERLINE): listed either way, re-evaluated by the command line with the
    # reference's operation sequence -- counted here because every one of them is host work outside this step
    n_borderline = None
    if lanes is None and calls_buf is not None and cap > 0:
        try:
            n_borderline = int((ctx.read_calls(dict(n_calls=n_calls, calls_buf=calls_buf, capacity=cap))["flags"] & 1).sum())
        except Exception:  # noqa: BLE001 -- a segment overflow is reported by the flags check, not here
            n_borderline = None
    # the whole poisson_call (stream + drain kernels back to back), outside the timed region
    ctx.set_async_drain(False)
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(5):
        ctx.poisson_call(tumours, P, fin.thr, ref_code, 100, mode=mode, call_mask=call_mask, capacity=cap, calls_buf=calls_buf, n_calls=n_calls)
    ctx.record(e1)
    t_call_sep = ctx.elapsed_ms(e0, e1) / 5
    # without --async-drain the whole call sits on the main stream and the in-region HIP events are the measurement
    t_call = t_call_sep if args.async_drain else t_call_main
    # validation mode, outside the timed region: all six scores of every record, as the reference evaluates them
    e0, e1 = ctx.event(), ctx.event()
    n_calls.zero_()
    ctx.record(e0)
    for _ in range(3):
        ctx.poisson_call(tumours, P, fin.thr, ref_code, 100, mode=POISSON_FULL, call_mask=call_mask)
    ctx.record(e1)
    t_call_full = ctx.elapsed_ms(e0, e1) / 3
    flags = int(fin.flags.item())
    if flags != 0:
        raise SystemExit("error_finalize reported an exactness-envelope violation")

    # the local part of a step with no exchange at all (reduce + finalize of the rank's own table + poisson_call), same
    # buffers: at N > 1 the difference to ms_per_step is the communication that the pipeline did not hide
    def local_step():
        f = ctx.error_estimate(normals, P, 0.002, 100, out=fins[0])
        ctx.poisson_call(tumours, P, f.thr, ref_code, 100, mode=mode, call_mask=call_mask, capacity=cap, calls_buf=calls_buf, n_calls=n_calls)

    def timed_passes(fn, reps):
        fn()
        torch.cuda.synchronize()
        a, b = ctx.event(), ctx.event()
        t0_ = time.perf_counter()
        ctx.record(a)
        for _ in range(reps):
            fn()
        ctx.record(b)
        ms = ctx.elapsed_ms(a, b) / reps
        return ms, (time.perf_counter() - t0_) / reps * 1e3

    t_local_ms = None
    coll_ms, coll_bytes, strong_base = None, None, None
    single_batch_ms = None
    if multi:
        # ONE batch with nothing else in flight: reduce -> exchange -> finalize of the slice -> all-gather -> poisson_call, host
        # clock around it, the slowest rank's figure.  What a single cohort through the command lines sees (their pipeline has
        # no other batch to hide the collectives behind); ms_per_step above is the pipelined throughput.
        lat = []
        for _ in range(5):
            fence()
            t1 = time.perf_counter()
            run_steps(1, False)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t1)
        tl = torch.tensor([sorted(lat)[len(lat) // 2]], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(tl, op=dist.ReduceOp.MAX)
        single_batch_ms = float(tl.item()) * 1e3
        materialise()
        if fins[0] is None:
            fins[0] = ctx.error_estimate(normals, P, 0.002, 100)
        t_local_ms, _ = timed_passes(local_step, max(10, args.steps))
        fence()
        if sliced:  # each collective of the exchange alone (one round serves G batches), nothing else on the devices
            coll_ms = merger.time_collectives(10)
            coll_bytes = merger.bytes_received_per_step()
            fence()
        # north_star words its scaling target on the TUMOUR SHARD ("tumour files sharded across the GPUs as embarrassingly-parallel work",
        # ">= 0.95 strong scaling of R_VC"): poisson_call of the rank's own T / N tumours against the finished table, no
        # exchange in it -- every rank times its shard (nothing else in flight), the slowest one counts
        tumour_shard = None
        if cfg["strong"]:
            fs = fins[0] if fins[0] is not None else fin

            def shard_call():
                ctx.poisson_call(tumours, P, fs.thr, ref_code, 100, mode=mode, call_mask=call_mask, capacity=cap, calls_buf=calls_buf, n_calls=n_calls)

            fence()
            # cold calls: the rank's own error_estimate between two of them, events around the call alone (ten calls back to back, as
            # round 5 timed it, replay a 128-tumour shard of 205 MB out of the 256 MiB Infinity Cache while the one-GPU base reads HBM)
            esh = [[ctx.event(), ctx.event()] for _ in range(10)]
            evict_sh = ctx.error_estimate(normals, P, 0.002, 100)
            for i in range(12):
                ctx.error_estimate(normals, P, 0.002, 100, out=evict_sh)
                if i >= 2:
                    ctx.record(esh[i - 2][0])
                shard_call()
                if i >= 2:
                    ctx.record(esh[i - 2][1])
            torch.cuda.synchronize()
            ms_sh = sorted(ctx.elapsed_ms(a, b) for a, b in esh)[5]
            tsh = torch.tensor([ms_sh], dtype=torch.float64, device=ctx.device)
            dist.all_reduce(tsh, op=dist.ReduceOp.MAX)
            tumour_shard = {"tumours_per_gpu": T, "poisson_call_ms_slowest_rank": float(tsh.item()), "R_VC_evals_per_s": P * T_total / (float(tsh.item()) * 1e-3)}
            fence()
        if cfg["strong"] and not args.no_strong_base:
            # The SAME job on ONE GPU, timed by rank 0 while the others wait: the base the driver's 1-GPU run cannot be (that
            # run is config 3, this job is config 4), so that every N > 1 line carries its own speed-up and efficiency.
            if rank == 0:
                del normals, tumours, packed
                torch.cuda.empty_cache()
                an = ctx.synth_fill(P, S_total, first_sample=0, seed=SEED, depth=depth)
                an = ctx.pack(an, layout)[0] if layout != "i32" else an
                at = ctx.synth_fill(P, T_total, first_sample=0, seed=SEED, depth=depth, tumour=True)
                at = ctx.pack(at, layout)[0] if layout != "i32" else at
                mask1 = torch.empty((T_total, P), dtype=torch.uint8, device=ctx.device)
                f1 = ctx.error_estimate(an, P, 0.002, 100)

                def whole_job():
                    ctx.error_estimate(an, P, 0.002, 100, out=f1)
                    ctx.poisson_call(at, P, f1.thr, ref_code, 100, mode=mode, call_mask=mask1, capacity=cap, calls_buf=calls_buf, n_calls=n_calls)

                ms1, _ = timed_passes(whole_job, 10)
                # ... and the calling half alone: all T tumours of the job on this one GPU (the base of tumour_shard)
                ms1_vc, _ = timed_passes(lambda: ctx.poisson_call(at, P, f1.thr, ref_code, 100, mode=mode, call_mask=mask1, capacity=cap, calls_buf=calls_buf,
                                                                  n_calls=n_calls), 10)
                if ctx.flags(clear=True) != 0:
                    raise SystemExit("kernel flags raised in the one-GPU base of the strong-scaling job")
                if tumour_shard:
                    tumour_shard.update({"one_gpu_poisson_call_ms": ms1_vc, "one_gpu_R_VC_evals_per_s": P * T_total / (ms1_vc * 1e-3),
                                         "speedup": ms1_vc / tumour_shard["poisson_call_ms_slowest_rank"],
                                         "efficiency": ms1_vc / tumour_shard["poisson_call_ms_slowest_rank"] / world})
                strong_base = {"n_gpus": 1, "ms_per_step": ms1, "value": (P * S_total + P * T_total) / (ms1 * 1e-3), "passes": 10,
                               "note": "the whole job (all normals, all tumours of the configuration) on rank 0's GPU alone, after the timed "
                                       "region, HIP events around 10 passes; same kernels, same record layout"}
                del an, at, mask1, f1
                normals = tumours = None
                packed = {}
            fence()
    # north_star's scaling target is worded on the TUMOUR SHARD, which has no exchange step -- so what one of N GPUs would run on config 4
    # (100 k positions x 1024 tumours, cut N ways along the tumour axis) can be timed on ONE GPU: poisson_call over 1024 / N tumours
    # against the 1024-tumour call / N, same box, same process, same table.  Every call is COLD: an error_estimate over this run's
    # normals (>= 400 MB) sits between two calls, so that a 128-tumour shard (205 MB) is not replayed out of the 256 MiB Infinity Cache.
    shard_proj = None
    if not multi and lanes is None and not args.no_shard_projection and mode == POISSON_PREFILTER and P == CONFIGS["c4"]["P"] and layout == "u16":
        T4 = CONFIGS["c4"]["T"]
        parts = []
        for lo in range(0, T4, 128):
            raw = ctx.synth_fill(P, 128, first_sample=lo, seed=SEED, depth=CONFIGS["c4"]["depth"], tumour=True)
            parts.append(ctx.pack(raw, layout)[0])
            del raw
        at = torch.cat(parts)
        del parts
        mask4 = torch.empty((T4, P), dtype=torch.uint8, device=ctx.device)
        f4 = fins[0] if fins[0] is not None else fin
        evict4 = ctx.error_estimate(normals, P, 0.002, 100)  # a table of its own: f4's thresholds stay what the calls read
        reps4 = 12

        def shard_ms(tn):
            e4 = [[ctx.event(), ctx.event()] for _ in range(reps4)]
            for i in range(reps4 + 2):
                ctx.error_estimate(normals, P, 0.002, 100, out=evict4)  # pushes the tumours out of the cache
                if i >= 2:
                    ctx.record(e4[i - 2][0])
                ctx.poisson_call(at[:tn], P, f4.thr, ref_code, 100, mode=mode, call_mask=mask4[:tn], capacity=cap, calls_buf=calls_buf, n_calls=n_calls)
                if i >= 2:
                    ctx.record(e4[i - 2][1])
            torch.cuda.synchronize()
            v = sorted(ctx.elapsed_ms(a, b) for a, b in e4)
            return v[len(v) // 2], v[0]

        def sequence_ms(tn):
            """(evicting error_estimate, then the call) x reps with TWO events in all: the pair's time, nothing timed in between"""
            ea4, eb4 = ctx.event(), ctx.event()
            for i in range(reps4 + 2):
                if i == 2:
                    ctx.record(ea4)
                ctx.error_estimate(normals, P, 0.002, 100, out=evict4)
                if tn:
                    ctx.poisson_call(at[:tn], P, f4.thr, ref_code, 100, mode=mode, call_mask=mask4[:tn], capacity=cap, calls_buf=calls_buf, n_calls=n_calls)
            ctx.record(eb4)
            torch.cuda.synchronize()
            return ctx.elapsed_ms(ea4, eb4) / reps4

        evict_only = sequence_ms(0)
        rows4, base4, base4s = [], None, None
        for n in (1, 2, 4, 8):
            med, mn = shard_ms(T4 // n)
            seq = sequence_ms(T4 // n) - evict_only  # what the call adds to a stream of work (no event bubbles around it)
            base4 = med if n == 1 else base4
            base4s = seq if n == 1 else base4s
            rows4.append({"n_gpus": n, "tumours_per_gpu": T4 // n, "poisson_call_ms": med, "min_ms": mn, "efficiency": base4 / n / med,
                          "poisson_call_ms_in_sequence": seq, "efficiency_in_sequence": base4s / n / seq,
                          "R_VC_evals_per_s_whole_job": P * T4 / (med * 1e-3)})
        if ctx.flags(clear=True) != 0:
            raise SystemExit("kernel flags raised in the tumour-shard projection")
        # t(T) = a + b T through the two end points: the fixed cost of a call and what it does to the eighth
        b4 = (rows4[0]["poisson_call_ms"] - rows4[3]["poisson_call_ms"]) / (T4 - T4 // 8)
        a4 = rows4[3]["poisson_call_ms"] - b4 * (T4 // 8)
        b4s = (rows4[0]["poisson_call_ms_in_sequence"] - rows4[3]["poisson_call_ms_in_sequence"]) / (T4 - T4 // 8)
        shard_proj = {"workload": CONFIGS["c4"]["name"], "records": layout, "passes_per_size": reps4, "by_n_gpus": rows4,
                      "efficiency_at_8": rows4[3]["efficiency"], "fixed_cost_ms": a4, "ms_per_tumour": b4,
                      "efficiency_at_8_in_sequence": rows4[3]["efficiency_in_sequence"],
                      "fixed_cost_ms_in_sequence": rows4[3]["poisson_call_ms_in_sequence"] - b4s * (T4 // 8), "evicting_error_estimate_ms": evict_only,
                      "note": "ONE GPU: poisson_call (stream kernel + drain kernel, HIP events, median) over config 4's first 1024 / N tumours against the same "
                              "call over all 1024, divided by N; every call cold (an error_estimate of this run's normals between two calls).  efficiency = "
                              "t(1024) / N / t(1024 / N).  fixed_cost_ms / ms_per_tumour: the line through the N = 1 and N = 8 points -- the drain's latency "
                              "chain, two launch gaps and the ramp of a launch do not shrink with the shard (DESIGN 7).  *_in_sequence: the same calls "
                              "timed as what they ADD to a stream of work -- (error_estimate, call) x passes between two events minus the "
                              "error_estimate alone timed the same way -- i.e. without the ~4 us bubble each bracketing event costs"}
        del at, mask4
        torch.cuda.empty_cache()
    sustained = None
    if not multi and lanes is None and args.sustained > 0:
        # the headline's timed window is a few milliseconds; this is the same pass repeated for ~half a second
        ms_dev, ms_host = timed_passes(local_step, args.sustained)
        sustained = {"passes": args.sustained, "ms_per_step": ms_dev, "value": (P * S + P * T) / (ms_dev * 1e-3),
                     "seconds": ms_host * args.sustained / 1e3, "note": "outside the contract's timed region; HIP events around the whole run"}
        if ctx.flags(clear=True) != 0:
            raise SystemExit("kernel flags raised in the sustained block")

    # The pass above is repeated over ONE resident batch (the contract's step), and the 256 MiB Infinity Cache keeps part of a
    # 400-800 MB cohort from one pass to the next.  The same pass rotated over three DISTINCT resident batches finds nothing of
    # its inputs in that cache: what a cohort streamed once from HBM sees.  Outside the timed region; reported beside `value`.
    cold = None
    if not multi and lanes is None and args.cold_batches > 1:
        extra = []
        for b in range(1, args.cold_batches):
            an = ctx.synth_fill(P, S, first_sample=b * 4096, seed=SEED, depth=depth)
            at = ctx.synth_fill(P, T, first_sample=b * 4096, seed=SEED, depth=depth, tumour=True)
            if layout != "i32":
                an, at = ctx.pack(an, layout)[0], ctx.pack(at, layout)[0]
            extra.append((an, at))
        sets = [(normals, tumours)] + extra
        reps = 10 * len(sets)
        cev = [[ctx.event() for _ in range(3)] for _ in range(reps)]

        def cold_pass(i, rec):
            an, at = sets[i % len(sets)]
            if rec:
                ctx.record(cev[i][0])
            f = ctx.error_estimate(an, P, 0.002, 100, out=fins[0])
            if rec:
                ctx.record(cev[i][1])
            ctx.poisson_call(at, P, f.thr, ref_code, 100, mode=mode, call_mask=call_mask, capacity=cap, calls_buf=calls_buf, n_calls=n_calls)
            if rec:
                ctx.record(cev[i][2])

        if fins[0] is None:
            fins[0] = ctx.error_estimate(normals, P, 0.002, 100)
        for i in range(len(sets)):
            cold_pass(i, False)
        torch.cuda.synchronize()
        for i in range(reps):
            cold_pass(i, True)
        torch.cuda.synchronize()
        c_red = sum(ctx.elapsed_ms(e[0], e[1]) for e in cev) / reps
        c_call = sum(ctx.elapsed_ms(e[1], e[2]) for e in cev) / reps
        c_step = ctx.elapsed_ms(cev[0][0], cev[-1][2]) / reps
        cold = {"batches": len(sets), "passes": reps, "ms_per_step": c_step, "value": (P * S + P * T) / (c_step * 1e-3),
                "error_reduce_ms": c_red, "error_reduce_frac_of_peak": (rec_bytes * P * S + 88 * P) / (c_red * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "poisson_call_ms": c_call, "poisson_call_frac_of_peak": (rec_bytes * P * T + 33 * P + P * T) / (c_call * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "note": "the same pass rotated over distinct resident batches (each pass reads records no earlier pass left in the 256 MiB Infinity "
                        "Cache); outside the contract's timed region, HIP events per pass"}
        cold_pass(0, False)  # leave the timed batch's own table and call mask in the output buffers (compared with below)
        torch.cuda.synchronize()
        if ctx.flags(clear=True) != 0:
            raise SystemExit("kernel flags raised in the cold-HBM block")
        del extra, sets

    # error_reduce against the SIZE of its launch: config 3 is 1563 tiles of 64 positions on 1280 (compact kernel) or 1024 resident
    # workgroups, i.e. one full round and a thin one; the same kernel on a panel of four whole rounds shows what the partly filled
    # round costs (tools/sweep_tiles.py).  Outside the timed region; N = 1 only.
    # the practical ceiling beside the spec: a plain device-to-device copy of 1 GiB on this box, read + written bytes per second
    copy_ceiling = None
    if not multi and lanes is None:
        try:
            nbytes = 1 << 30
            src_c, dst_c = torch.empty(nbytes, dtype=torch.uint8, device=ctx.device), torch.empty(nbytes, dtype=torch.uint8, device=ctx.device)
            src_c.zero_()
            dst_c.copy_(src_c)
            torch.cuda.synchronize()
            ca, cb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ca.record()
            for _ in range(10):
                dst_c.copy_(src_c)
            cb.record()
            torch.cuda.synchronize()
            copy_ms = ca.elapsed_time(cb) / 10
            copy_ceiling = {"GBs": 2 * nbytes / (copy_ms * 1e-3) / 1e9, "bytes_copied": nbytes, "ms": copy_ms,
                            "note": "torch device-to-device copy of 1 GiB, read + written bytes, 10 copies back to back, after the timed region"}
            del src_c, dst_c
        except Exception as exc:  # noqa: BLE001
            copy_ceiling = {"error": f"{type(exc).__name__}: {exc}"}
    whole_rounds = None
    if not multi and lanes is None and args.whole_rounds > 0:
        try:
            kname = ctx.last_reduce_kernel()
            # workgroups a CU holds: 96 VGPRs -> five waves per SIMD (compact kernel), 120 -> four (general); one wave of each of a CU's SIMDs per workgroup
            per_round = torch.cuda.get_device_properties(dev_index).multi_processor_count * (5 if kname in ("error_reduce_u16_kernel", "error_reduce_u24_kernel") else 4)
            P2 = per_round * args.whole_rounds * 64
            an2 = ctx.synth_fill(P2, S, seed=SEED, depth=depth)
            if layout != "i32":
                an2 = ctx.pack(an2, layout)[0]
            f2 = ctx.error_estimate(an2, P2, 0.002, 100)
            for _ in range(2):
                ctx.error_estimate(an2, P2, 0.002, 100, out=f2)
            ea, eb = ctx.event(), ctx.event()
            ctx.record(ea)
            for _ in range(10):
                ctx.error_estimate(an2, P2, 0.002, 100, out=f2)
            ctx.record(eb)
            torch.cuda.synchronize()
            t2 = ctx.elapsed_ms(ea, eb) / 10
            b2 = rec_bytes * P2 * S + 88 * P2
            whole_rounds = {"kernel": ctx.last_reduce_kernel(), "positions": P2, "tiles": P2 // 64, "resident_workgroups": per_round,
                            "rounds_of_workgroups": args.whole_rounds, "config_rounds_of_workgroups": ((P + 63) // 64) / per_round,
                            "avg_ms": t2, "algorithmic_bytes": b2, "achieved": b2 / (t2 * 1e-3) / 1e9, "unit": "GB/s",
                            "frac": b2 / (t2 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            "note": "the dominant kernel on a panel that is a whole number of rounds of resident workgroups (same samples per position, "
                                    "same layout), 10 launches back to back, HIP events; roofline.frac is the configuration's own launch"}
            if ctx.flags(clear=True) != 0:
                raise SystemExit("kernel flags raised in the whole-rounds block")
            del an2, f2
        except (Exception, SystemExit) as exc:  # an extra block must not cost the run its line: the failure is reported in its place
            whole_rounds = {"error": f"{type(exc).__name__}: {exc}"}
            print(f"bench.py: the whole_rounds block failed: {exc}", file=sys.stderr)

    # With position ranges in the timed region, the same pass once more WITHOUT them: every launch whole, on one stream, HIP events
    # around every kernel of every pass -- what rounds 1-4 timed, and the undisturbed kernel durations beside the overlapped ones.
    one_stream = None
    if n_ranges > 1:
        try:
            ev1 = [[ctx.event() for _ in range(3)] for _ in range(args.steps)]
            for _ in range(max(2, args.warmup)):
                local_step()
            torch.cuda.synchronize()
            f1 = fins[0]
            for i in range(args.steps):
                ctx.record(ev1[i][0])
                ctx.error_estimate(normals, P, 0.002, 100, out=f1)
                ctx.record(ev1[i][1])
                ctx.poisson_call(tumours, P, f1.thr, ref_code, 100, mode=mode, call_mask=call_mask, capacity=cap, calls_buf=calls_buf, n_calls=n_calls)
                ctx.record(ev1[i][2])
            torch.cuda.synchronize()
            ms1 = ctx.elapsed_ms(ev1[0][0], ev1[-1][2]) / args.steps
            red1 = sum(ctx.elapsed_ms(e[0], e[1]) for e in ev1) / args.steps
            call1 = sum(ctx.elapsed_ms(e[1], e[2]) for e in ev1) / args.steps
            rb1, cb1 = rec_bytes * P * S + 88 * P, rec_bytes * P * T + 33 * P + P * T
            # the same passes once more with two events in all: what the one-stream pass costs when nobody times its kernels
            ea1, eb1 = ctx.event(), ctx.event()
            ctx.record(ea1)
            for _ in range(args.steps):
                local_step()
            ctx.record(eb1)
            torch.cuda.synchronize()
            ms1_plain = ctx.elapsed_ms(ea1, eb1) / args.steps
            same1 = (torch.equal(ranged_out[0].view(torch.int32), f1.thr.view(torch.int32)) and torch.equal(ranged_out[1], f1.code) and
                     torch.equal(ranged_out[2], f1.germ_present) and torch.equal(ranged_out[3], call_mask) and
                     ranged_out[4] == int(n_calls[::CALL_COUNTER_STRIDE].sum().item()))
            del ranged_out
            one_stream = {"steps": args.steps, "ms_per_step": ms1, "same_outputs_as_the_timed_region": bool(same1), "value": (P * S + P * T) / (ms1 * 1e-3), "kernel": ctx.last_reduce_kernel(),
                          "ms_per_step_without_kernel_events": ms1_plain, "value_without_kernel_events": (P * S + P * T) / (ms1_plain * 1e-3),
                          "error_reduce_ms": red1, "error_reduce_frac_of_peak": rb1 / (red1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "poisson_call_ms": call1, "poisson_call_frac_of_peak": cb1 / (call1 * 1e-3) / 1e9 / HBM_PEAK_GBS,
                          "note": "the pass of the timed region without position ranges (ampli_set_ranges(1)): whole launches on one stream, HIP events around "
                                  "every kernel of every pass (the events cost a few us per pass: ms_per_step_without_kernel_events = the same passes with two events in all); after the timed region"}
            if ctx.flags(clear=True) != 0:
                raise SystemExit("kernel flags raised in the one-stream block")
        except (Exception, SystemExit) as exc:  # an extra block must not cost the run its line: the failure is reported in its place
            one_stream = {"error": f"{type(exc).__name__}: {exc}"}
            print(f"bench.py: the one_stream block failed: {exc}", file=sys.stderr)

    others = []
    for name in [n for n in ("i32", "u24", "u16") if n in packed and n != layout and not multi and lanes is None]:
        # the same workload in the other record layouts, outside the timed region, for comparison
        an, at = packed[name]
        c2 = Context(dev_index)
        c2.set_record_layout(name)
        f2 = c2.error_estimate(an, P, 0.002, 100)
        r2 = c2.poisson_call(at, P, f2.thr, ref_code, 100, mode=mode, capacity=cap)
        same = all(torch.equal(getattr(f2, k).view(torch.uint8), getattr(fin, k).view(torch.uint8)) for k in ("rate", "thr", "code", "germ_present"))
        same = same and torch.equal(r2["call_mask"], call_mask)

        def avg_ms(fn, reps):
            fn()
            a, b = c2.event(), c2.event()
            c2.record(a)
            for _ in range(reps):
                fn()
            c2.record(b)
            return c2.elapsed_ms(a, b) / reps

        def step2():
            c2.error_estimate(an, P, 0.002, 100, out=f2)
            c2.poisson_call(at, P, f2.thr, ref_code, 100, mode=mode, call_mask=r2["call_mask"], capacity=cap,
                            calls_buf=r2["calls_buf"], n_calls=r2["n_calls"])

        ob = REC_BYTES[name]
        t2_red = avg_ms(lambda: c2.error_estimate(an, P, 0.002, 100, out=f2), 20)
        t2_step = avg_ms(step2, 20)
        others.append({"records": f"{name} ({ob} B per record)", "ms_per_step": t2_step, "value": (P * S + P * T) / (t2_step * 1e-3),
                       "error_reduce_ms": t2_red, "error_reduce_GBs": (ob * P * S + 88 * P) / (t2_red * 1e-3) / 1e9,
                       "error_reduce_frac_of_peak": (ob * P * S + 88 * P) / (t2_red * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "same_outputs": bool(same), "note": "20 passes, HIP events, outside the timed region"})
        c2.close()
        del an, at, f2, r2

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        evals = (P * S_total + P * T_total) * args.steps
        acc_bytes = ctx.lib.ampli_acc_bytes(P)
        # algorithmic bytes of error_reduce per launch: the records + what it writes (the accumulator table,
        # or at N = 1 the finalised error table: rate 32 B + thr 32 B + code 4 B + germ 16+4 B per position)
        # (N > 1, sliced merge: 21 -- or 14, slim format -- doubles + 8 floats per position into the exchange buffers = 200 / 144 B)
        red_bytes = rec_bytes * P * S + (((8 * merger.planes + 32) * P if sliced else acc_bytes) if multi else 88 * P)
        call_bytes = rec_bytes * P * T + 33 * P + P * T    # poisson_call: records + thresholds/ref + mask
        lay = {"i32": 0, "u16": 1, "u24": 2}[layout]
        # the library says which kernel its latest error_reduce launch was (uint16 records without positions listed twice go
        # through the compact-state kernel: five waves per SIMD; csrc/ampli_kernels.hip)
        red_name = ctx.last_reduce_kernel()
        if red_name == "error_reduce_kernel":
            red_name = f"error_reduce_kernel<true, 1, {lay}>"
        # With position ranges in the timed region a kernel is n launches per pass on n streams, each over 1 / n of the panel and sharing
        # the chip with the other ranges' kernels: such a launch's duration is not the kernel's (the fractions of concurrent kernels add
        # up: roofline_pass).  The roofline of the dominant KERNEL is therefore taken on whole, undisturbed launches: the one_stream block,
        # i.e. the same passes of this same process right behind the timed region, ranges off, HIP events around every launch -- what
        # `rocprofv3 --stats -- python bench.py --ranges 1` averages.  The overlapped launches are in roofline_overlapped / `ranges`.
        pass_red_bytes, pass_call_bytes = red_bytes, call_bytes
        overlapped = None
        roof_measured = "HIP events on the launch stream over the timed region"
        if n_ranges > 1:
            overlapped = {"launches_per_step": n_ranges, "error_reduce": {"avg_ms": t_red, "algorithmic_bytes": red_bytes / n_ranges,
                                                                          "frac": red_bytes / n_ranges / (t_red * 1e-3) / 1e9 / HBM_PEAK_GBS},
                          "poisson_call": {"avg_ms": t_call, "algorithmic_bytes": call_bytes / n_ranges,
                                           "frac": call_bytes / n_ranges / (t_call * 1e-3) / 1e9 / HBM_PEAK_GBS},
                          "note": f"the timed region's own launches: {n_ranges} per kernel and pass on {n_ranges} streams (ampli_set_ranges), events on each range's stream; "
                                  "a launch shares the chip with the other ranges' kernels, so these fractions are shares, not kernel efficiencies"}
            if one_stream and "error_reduce_ms" in one_stream:
                t_red, t_call = one_stream["error_reduce_ms"], one_stream["poisson_call_ms"]
                roof_measured = ("HIP events around every launch of the one_stream block: the timed region's passes repeated right behind it with "
                                 "position ranges off (whole launches on one stream); the timed region itself runs the kernels as overlapping range "
                                 "launches (roofline_overlapped, roofline_pass)")
            else:  # no undisturbed measurement: the overlapped launches it is
                red_bytes, call_bytes = red_bytes / n_ranges, call_bytes / n_ranges
                roof_measured = "HIP events on each range's stream over the timed region (overlapping launches; the one_stream block failed)"
        if t_red >= t_call:
            dom, dom_ms, dom_bytes = red_name, t_red, red_bytes
        else:
            dom, dom_ms, dom_bytes = f"poisson_stream_kernel<{lay}, false>+poisson_drain_kernel", t_call, call_bytes
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel from the PMC counters: collected with rocprofv3 --pmc in separate passes of
        # this same command (tools/collect_profiles.sh) and corrected as MI355X_MICROARCH.md prescribes; committed
        # under profiles/.  null when the workload is not the profiled one.
        traffic, traffic_src = None, None
        import glob

        pj = next(iter(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "pmc_summary.json")), reverse=True)), "")  # the latest round's
        if args.config == "c3" and not multi and os.path.exists(pj):
            try:
                pm = json.load(open(pj))
                if pm.get("kernel_source_sha256") == kernel_source_sha():  # offline counters of exactly these kernels, else nothing
                    traffic = sum(v.get("hbm_bytes_per_launch", 0.0) for k, v in pm["kernels"].items()
                                  if any(part in k for part in dom.split("+"))) or None
                    traffic_src = (os.path.relpath(pj, ROOT) + ": rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of this command on "
                                   "this kernel source (sha256 matches), gfx950 x2 read correction; offline, not collected in this run")
            except Exception:
                traffic = None
        # the other half of the step gets its own roofline entry (same definitions): algorithmic bytes, HIP-event time, PMC traffic
        if t_red >= t_call:
            oth, oth_ms, oth_bytes = f"poisson_stream_kernel<{lay}, false>+poisson_drain_kernel", t_call, call_bytes
        else:
            oth, oth_ms, oth_bytes = red_name, t_red, red_bytes
        oth_traffic = None
        if traffic_src:
            try:
                oth_traffic = sum(v.get("hbm_bytes_per_launch", 0.0) for k, v in pm["kernels"].items() if any(part in k for part in oth.split("+"))) or None
            except Exception:
                oth_traffic = None
        out = {
            "metric": "position-evaluations/s (error-est + Poisson call)",
            "value": evals / elapsed,
            "unit": "position-evaluations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "none" if world == 1 else ("strong" if cfg["strong"] else "weak"),
            "vs_baseline": None,
            "dtype": {"i32": "int32", "u24": "24-bit", "u16": "uint16"}[layout] + " counts; f64 sums / Poisson; f32 rates",
            "data": "synthetic",
            "config": {"workload": cfg["name"], "positions": P, "normals_per_gpu": S, "tumours_per_gpu": T, "normals_total": S_total, "tumours_total": T_total, "depth": depth,
                       "C_value": 0.002, "coverage_cutoff": 100, "poisson_mode": args.mode, "streams": args.streams if not multi else 1,
                       "position_ranges": n_ranges, **({"position_ranges_fallback": ranges_fallback} if ranges_fallback else {}),
                       "parallelism": f"tumour+normal sample shards x{world} ({'the fixed job split' if cfg['strong'] else 'one shard of the configuration per GPU'})" + (("; per batch: RCCL reduce-scatter of the sums + all-to-all of the germ-max pairs by position slice, finalize of the own slice, all-gather of the error table -- one round of collectives per group of independent batches, three groups in flight" if sliced else "; one packed RCCL all-reduce + all-gather of the germ-max regions per batch, overlapped with the neighbouring batches") if multi else ""),
                       "merge": (args.merge if multi else None), "batches_per_exchange": (G if sliced else None),
                       "rehearsal": ("N>1 code path forced on one rank (--force-dist)" if args.force_dist and world == 1 else None),
                       "records": f"{layout} ({rec_bytes} B per record: 8 fields x {rec_bytes} bits)",
                       "records_why": ("--records auto: the narrowest layout this workload's counts fit, which is the layout the command lines' host "
                                       "packer uploads for the same cohort (csrc/host/aseq.cpp; e2e block: error_estimation.record_array_MB)"
                                       if args.records == "auto" else "--records given")},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_ms": dom_ms, "algorithmic_bytes": dom_bytes,
                         "traffic_source": traffic_src, "measured": roof_measured,
                         "frac_of_copy_ceiling": (achieved / copy_ceiling["GBs"]) if copy_ceiling and copy_ceiling.get("GBs") else None},
            "roofline_overlapped": overlapped,
            # every kernel of a pass together: algorithmic bytes of the pass / time per pass over the timed region -- the figure that
            # is well defined when kernels of different ranges overlap
            "copy_ceiling": copy_ceiling,
            "roofline_pass": {"bound": "hbm", "algorithmic_bytes": pass_red_bytes + pass_call_bytes, "ms": ms_per_step,
                              "achieved": (pass_red_bytes + pass_call_bytes) / (ms_per_step * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": (pass_red_bytes + pass_call_bytes) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
            "roofline_other_kernel": {"bound": "hbm", "kernel": oth, "achieved": oth_bytes / (oth_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                      "frac": oth_bytes / (oth_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": oth_traffic, "avg_ms": oth_ms,
                                      "algorithmic_bytes": oth_bytes},
            "kernels": {"error_reduce_ms": t_red, "error_reduce_GBs": red_bytes / (t_red * 1e-3) / 1e9,
                        "poisson_call_ms": t_call, "poisson_call_GBs": call_bytes / (t_call * 1e-3) / 1e9,
                        "poisson_call_main_stream_ms": t_call_main, "drain": "side stream, overlapped with the next batch" if args.async_drain else "main stream",
                        "poisson_call_full_mode_ms": t_call_full,
                        # SURVEY.md 8d's three rates: panel positions/s through error estimation, tumour position-evaluations/s
                        # through calling, and tumour position-evaluations/s through the whole pass
                        "R_EE_positions_per_s": P / (t_red * 1e-3), "R_VC_evals_per_s": P * T / (t_call * 1e-3),
                        "R_pipe_evals_per_s": P * T_total / (ms_per_step * 1e-3)},
            "calls_per_step": n_found,
            "borderline_per_step": n_borderline,
        }
        # The all-scores mode (every record's six Poisson scores, as the reference evaluates them; outside the timed region) is
        # bound by FP64 vector issue, not by HBM: its own roofline, from the FP64 instruction counters of profiles/ (same
        # kernel source only).  achieved counts every lane of an issued FP64 wave instruction (an upper bound of the useful
        # flops: lanes that are masked off in the divergent series / continued-fraction loops are counted too), so
        # issue_slot_frac -- FP64 wave instructions x 4 cycles / (SIMDs x clock x time) -- is the figure to read.
        try:
            fk = next((v for k, v in pm["kernels"].items() if f"poisson_full_kernel<{lay}>" in k and "fp64_wave_instructions" in v), None) if traffic_src else None
        except Exception:  # noqa: BLE001
            fk = None
        out["roofline_fp64"] = {"bound": "fp64 vector issue", "kernel": f"poisson_full_kernel<{lay}> (all six scores of every record)", "avg_ms": t_call_full,
                                "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                                "achieved": (fk["fp64_flops_if_all_lanes_active"] / (t_call_full * 1e-3) / 1e12) if fk else None,
                                "frac": (fk["fp64_flops_if_all_lanes_active"] / (t_call_full * 1e-3) / 1e12 / FP64_PEAK_TFLOPS) if fk else None,
                                "fp64_wave_instructions_per_launch": fk["fp64_wave_instructions"] if fk else None,
                                "issue_slot_frac": (fk["fp64_wave_instructions"] * 4 / (1024 * 2.4e9 * t_call_full * 1e-3)) if fk else None,
                                "hbm_frac_of_the_same_launch": call_bytes / (t_call_full * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "source": "SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 of profiles/*/pmc_summary.json (offline, same kernel source)" if fk else
                                          "no FP64 counters for this kernel source under profiles/ (run tools/collect_profiles.sh)"}
        if others:
            out["other_record_layouts"] = others
        if sustained:
            out["sustained"] = sustained
        if shard_proj:
            out["tumour_shard_projection"] = shard_proj
        if cold:
            out["cold_hbm"] = cold
        if whole_rounds:
            out["roofline_whole_rounds"] = whole_rounds
        if ranges_block:
            out["ranges"] = ranges_block
        if one_stream:
            out["one_stream"] = one_stream
        if multi:
            out["communication"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "merge": args.merge,
                                    "local_step_ms": t_local_ms, "exposed_ms_per_step": max(0.0, ms_per_step - t_local_ms),
                                    "bytes_received_per_rank_and_step": coll_bytes,
                                    "collective_ms_per_round": coll_ms, "batches_per_round": (G if sliced else 1),
                                    "single_batch_latency_ms": single_batch_ms,
                                    "single_batch_value": (P * S_total + P * T_total) / (single_batch_ms * 1e-3) if single_batch_ms else None,
                                    "unverified_on_more_than_one_gpu": "efficiency / strong_base / exposed_ms have only been exercised with one rank over RCCL or gloo ranks sharing a GPU until a multi-GPU SCALE run exists",
                                    "note": "single_batch_latency_ms = one batch with nothing else in flight (median of 5, slowest rank, host clock): the "
                                            "unpipelined latency beside the pipelined ms_per_step; local_step = reduce + finalize of the rank's own table + poisson_call with no exchange (rank 0, after the "
                                            "timed region); exposed = ms_per_step - local_step; collective_ms_per_round: each collective alone, 10 rounds "
                                            "back to back after the timed region (one round serves batches_per_round batches); world_size and backend as "
                                            "torch.distributed reports them (nccl = RCCL)"}
            if tumour_shard:
                tumour_shard["note"] = ("strong scaling of the tumour shard alone (R_VC, no exchange step): poisson_call of each rank's T / N tumours, median of 10 cold "
                                        "calls (the rank's own error_estimate between two of them), the slowest rank's time, against the same call over all T tumours on rank "
                                        "0's GPU alone; unverified on more than one GPU until a SCALE run exists")
                out["tumour_shard"] = tumour_shard
                out["tumour_shard_efficiency"] = tumour_shard.get("efficiency")
            if strong_base and single_batch_ms:
                # one cohort, nothing to pipeline against: the unpipelined latency of a batch against the same job on one GPU
                out["single_cohort_efficiency"] = strong_base["ms_per_step"] / single_batch_ms / world
            if strong_base:
                out["strong_base"] = strong_base
                out["strong_base_value"] = strong_base["value"]
                out["speedup_vs_one_gpu"] = out["value"] / strong_base["value"]
                out["efficiency"] = out["value"] / strong_base["value"] / world
        if not multi and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg, fin.thr.cpu().numpy(), ref_code.cpu().numpy(), full=args.cpu_baseline_full)
            except Exception as e:  # the baseline leg must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "position-evaluations/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        if not multi and not args.no_e2e:
            # end to end through the two executables: the small configuration whole (reference on the same files, tables
            # compared byte for byte) and this workload (>= 1 GB of ASEQ text at config 3; reference on a subset)
            try:
                del normals, tumours, packed
                torch.cuda.empty_cache()
                out["e2e"] = {"c2": e2e_leg("c2", CONFIGS["c2"]), args.config: e2e_leg(args.config, cfg, ref_files=6)}
            except Exception as e:
                out["e2e"] = {"error": repr(e)}
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
