#!/usr/bin/env python3
"""Where a command line's wall time goes: both executables on BASELINE config 2 and config 3 written out as ASEQ text,
each run `--reps` times with AMPLISOLVE_TIMING=1; prints the outer wall clock and the TIMING2 phase lines (critical spans
add up to the time inside main(); overlapped spans ran on other threads).  usage: python tools/cli_phases.py [--reps 3] [--configs c2,c3]"""
import argparse
import json
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def phases(err):
    out = {}
    for ln in err.splitlines():
        if ln.startswith("TIMING2 "):
            w = ln.split()
            out[w[1] + ("" if w[3] != "overlapped" else "*")] = round(float(w[2]), 4)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--configs", default="c2,c3")
    ap.add_argument("--env", default="", help="extra environment, K=V,K=V")
    a = ap.parse_args()
    extra = dict(kv.split("=", 1) for kv in a.env.split(",") if kv)
    for name in a.configs.split(","):
        cfg = bench.CONFIGS[name]
        d = tempfile.mkdtemp(prefix=f"ampli_phases_{name}_")
        try:
            bench.write_workload_files(d, cfg["P"], cfg["S"], cfg["T"], cfg["depth"])
            env = {"AMPLISOLVE_TIMING": "1", "AMPLISOLVE_STRICT_EXIT": "1", "AMPLISOLVE_REFBASES_FILE": "refbases.txt", **extra}
            for rep in range(a.reps):
                rc, wall, _, out, err, rss = bench._run_timed([os.path.join(bench.BIN, "AmpliSolveErrorEstimation"), "panel_design=panel.bed", "reference_genome=unused.fa",
                                                                "germline_dir=N", "C_value=0.002", "coverage_cutoff=100", "default_error=0.01", "output_dir=ee"], d, env)
                assert rc == 0, out + err
                print(json.dumps({"config": name, "exe": "EE", "rep": rep, "wall_s": round(wall, 4), "phases": phases(err)}), flush=True)
                rc, wall, _, out, err, rss = bench._run_timed([os.path.join(bench.BIN, "AmpliSolveVariantCalling"), "errorFile=ee/positionSpecificNoise_0.0020.txt",
                                                                "tumour_dir=T", "output_dir=vc", "coverage_cutoff=100", "p_value=0.05"], d, env)
                assert rc == 0, out + err
                print(json.dumps({"config": name, "exe": "VC", "rep": rep, "wall_s": round(wall, 4), "phases": phases(err)}), flush=True)
        finally:
            shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    main()
