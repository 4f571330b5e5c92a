"""Freshly drawn panels through the REFERENCE's own error-estimation code (oracle/_ref/ee_ref_driver, compiled from the reference
sources where they lie) and through the drop-in command line, in the same directory with the same literals: the two error tables
must be the same bytes.  Random C / coverage cut-off / depth (depths that make the host packer take uint16, 24-bit and int32 records),
sample counts, chunk sizes (one sample per chunk ... everything in one) and parser thread counts.
Usage (GPU box, /root/reference not needed: the driver is a prebuilt file): python tools/fuzz_cli_vs_reference.py [cases] [first seed]"""
import os
import pathlib
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import pyoracle as orc
from tests.helpers import write_fresh_panel

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "amplisolve_amd", "bin")
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
assert os.path.exists(orc.REF_EE_DRIVER), "oracle/_ref/ee_ref_driver is absent (make -C oracle where /root/reference exists)"
bad = 0
inorder = 0
for case in range(n_cases):
    seed = seed0 + case
    rng = np.random.default_rng(seed)
    depth = int(rng.choice([300, 2000, 6000, 30_000, 120_000, 400_000, 3_000_000, 40_000_000]))
    S = int(rng.integers(3, 24))
    C = f"{float(rng.choice([0.0005, 0.002, 0.01, 0.03])):g}"
    cov = str(int(rng.choice([1, 30, 100, 400, 2000])))
    env = dict(os.environ, AMPLISOLVE_STRICT_EXIT="1", AMPLISOLVE_REFBASES_FILE="r.txt", AMPLISOLVE_TIMING="1",
               AMPLISOLVE_THREADS=str(int(rng.integers(1, 9))))
    if rng.random() < 0.7:
        env["AMPLISOLVE_CHUNK_BYTES"] = str(int(rng.choice([1, 40_000, 200_000, 2_000_000])))
    with tempfile.TemporaryDirectory(prefix="ampli_fuzz_") as td:
        d = pathlib.Path(td)
        write_fresh_panel(d, seed, depth=depth, S=S, amplicons=int(rng.integers(3, 10)))
        (d / "o").mkdir()
        ref = subprocess.run([orc.REF_EE_DRIVER, "p.bed", "r.txt", "d.txt", "N", C, cov, "o"], capture_output=True, text=True, cwd=d)
        ours = subprocess.run([f"{BIN}/AmpliSolveErrorEstimation", "panel_design=p.bed", "reference_genome=x.fa", "germline_dir=N", f"C_value={C}",
                               f"coverage_cutoff={cov}", "default_error=0.01", "output_dir=q"], capture_output=True, text=True, cwd=d, env=env)
        names = [n for n in os.listdir(d / "o") if n.startswith("positionSpecificNoise_")] if ref.returncode == 0 else []
        ok = ref.returncode == 0 and ours.returncode == 0 and len(names) == 1 and (d / "o" / names[0]).read_bytes() == (d / "q" / names[0]).read_bytes()
        t = [ln.split() for ln in ours.stderr.splitlines() if ln.startswith("TIMING stream")]
        per_rec = None
        if t:
            mb, lines = float(t[0][t[0].index("record_MB") + 1]), float(t[0][t[0].index("lines") + 1])
            chunks = int(t[0][t[0].index("chunks") + 1])
        size = (d / "o" / names[0]).stat().st_size if names else 0
        reran = "in-order pass" if "summing again in the reference's order" in ours.stdout else ""
        print(f"seed {seed}: depth {depth:>9} S {S:2d} C {C:>6} cov {cov:>4} threads {env['AMPLISOLVE_THREADS']} chunk_bytes {env.get('AMPLISOLVE_CHUNK_BYTES', 'default'):>8} "
              f"chunks {chunks if t else '?':>3} record_MB {mb if t else -1:8.3f}  table {size:7d} B  {'IDENTICAL' if ok else 'NOT IDENTICAL'} {reran}", flush=True)
        if not ok:
            bad += 1
            msg = [ln.strip() for ln in (ours.stdout + ours.stderr).splitlines() if "envelope" in ln.lower() or "something went wrong" in ln.lower()]
            print("   reference rc", ref.returncode, "ours rc", ours.returncode, "; ".join(msg)[:600] or (ours.stdout + ours.stderr)[-600:], flush=True)
        elif "summing again in the reference's order" in ours.stdout:
            inorder += 1
print(f"{n_cases} cases, {bad} different ({inorder} of the identical ones outside the exactness envelope: summed again in the reference's order, DESIGN 4.2)")
sys.exit(1 if bad else 0)
