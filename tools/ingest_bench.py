"""Host ingest + PCIe-inclusive rate of the C++ host (f1): writes a synthetic cohort as .PILEUP.ASEQ text, then times
 (a) the C++ parser/packer (mmap + threads) and (b) the whole AmpliSolveErrorEstimation run (parse + H2D + kernels + write)."""
import os, subprocess, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np
from bench import synthetic_panel, _write_aseq_dir
from tests.helpers import synth_recs, synth_ref
from amplisolve_amd.hostio import HostCohort

P, S = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, int(sys.argv[2]) if len(sys.argv) > 2 else 64
rows, pos = synthetic_panel(P)
recs = synth_recs(P, S); refb = synth_ref(P)
d = tempfile.mkdtemp(prefix="ampli_ingest_")
t = time.time(); _write_aseq_dir(recs, pos, f"{d}/normals"); print(f"wrote {S} files in {time.time()-t:.1f}s")
open(f"{d}/panel.bed", "w").write("".join(f"{c}\t{a}\t{b}\tA{i}\trs{i}\tG{i}\n" for i, (c, a, b) in enumerate(rows)))
open(f"{d}/refbases.txt", "w").write("".join(f"{c}\t{p}\t{'ACGT'[refb[i]]}\n" for i, (c, p) in enumerate(pos)))
size = sum(os.path.getsize(f"{d}/normals/{f}") for f in os.listdir(f"{d}/normals"))
for th in (1, 4, 16):
    t = time.time(); co = HostCohort(f"{d}/panel.bed", f"{d}/normals", refbases_file=f"{d}/refbases.txt", threads=th); dt = time.time() - t
    st = co.stats(); print(f"parse+pack threads={th}: {dt:.3f}s  {st['lines']/dt/1e6:.2f} M lines/s  {size/dt/1e9:.2f} GB/s text"); co.close()
env = dict(os.environ, AMPLISOLVE_REFBASES_FILE=f"{d}/refbases.txt", AMPLISOLVE_TIMING="1", AMPLISOLVE_STRICT_EXIT="1")
t = time.time()
r = subprocess.run(["amplisolve_amd/bin/AmpliSolveErrorEstimation", f"panel_design={d}/panel.bed", "reference_genome=x.fa", f"germline_dir={d}/normals",
                    "C_value=0.002", "coverage_cutoff=100", "default_error=0.01", f"output_dir={d}/out"], capture_output=True, text=True, env=env)
print("CLI end-to-end", round(time.time() - t, 3), "s rc", r.returncode); print(r.stderr.strip())
