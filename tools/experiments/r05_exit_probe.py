"""ON THE GPU BOX: what a config-4-sized AmpliSolveErrorEstimation run spends AFTER main()'s report (0.09-0.14 s against 0.002 s
at config 3): the same command on the same files, repeated, with the ring left unpinned or smaller, with four parser threads, with the orderly exit.  Prints wall / in main / before main / after report per run."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench

cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c4"]
d = tempfile.mkdtemp(prefix="ampli_exit_probe_")
bench.write_workload_files(d, cfg["P"], cfg["S"], cfg["T"], cfg["depth"])
cmd = [os.path.join(bench.BIN, "AmpliSolveErrorEstimation"), "panel_design=panel.bed", "reference_genome=unused.fa", "germline_dir=N", "C_value=0.002",
       "coverage_cutoff=100", "default_error=0.01", "output_dir=ee"]
base = {"AMPLISOLVE_TIMING": "1", "AMPLISOLVE_STRICT_EXIT": "1", "AMPLISOLVE_REFBASES_FILE": "refbases.txt"}


def run(tag, extra=None):
    rc, wall, t, out, err, rss = bench._run_timed(cmd, d, dict(base, **(extra or {})))
    print(f"{tag:28s} rc {rc} wall {wall:.3f} in_main {t.get('phases', {}).get('wall_in_main')} outside {t.get('outside_main')} rss_MB {rss:.0f}", flush=True)


subprocess.run(["sync"])
for rep in range(3):
    run(f"run {rep + 1}")
run("unpinned ring", {"AMPLISOLVE_PIN": "none"})
run("ring of 512 MB", {"AMPLISOLVE_RING_MB": "512"})
run("4 parser threads", {"AMPLISOLVE_THREADS": "4"})
run("orderly exit", {"AMPLISOLVE_EXIT": "orderly"})
subprocess.run(["rm", "-rf", d])
