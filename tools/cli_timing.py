"""End-to-end wall time of the two command lines on the staged full Toy_data (tests/golden/_toy_full), GPU box."""
import gzip, os, subprocess, sys, tempfile, time
T = "tests/golden/_toy_full"
if not os.path.isdir(T):
    sys.exit("no staged Toy_data")
O = tempfile.mkdtemp()
open(f"{O}/ref.txt", "wb").write(gzip.open("tests/golden/toy/refbases.txt.gz").read())
env = dict(os.environ, AMPLISOLVE_REFBASES_FILE=f"{O}/ref.txt", AMPLISOLVE_TIMING="1")
for i in range(3):
    t0 = time.time()
    r = subprocess.run(["amplisolve_amd/bin/AmpliSolveErrorEstimation", f"panel_design={T}/AmpliSeq_30genes_Designed-1.bed", "reference_genome=x.fa",
                        f"germline_dir={T}/NORMAL_ASEQ_DIR", "C_value=0.002", "coverage_cutoff=100", "default_error=0.01", f"output_dir={O}/ee"],
                       capture_output=True, text=True, env=env)
    t1 = time.time()
    v = subprocess.run(["amplisolve_amd/bin/AmpliSolveVariantCalling", f"errorFile={O}/ee/positionSpecificNoise_0.0020.txt", f"tumour_dir={T}/TUMOUR_ASEQ_DIR",
                        f"output_dir={O}/vc", "coverage_cutoff=100", "p_value=0.05"], capture_output=True, text=True, env=env)
    t2 = time.time()
    print(f"run {i}: AmpliSolveErrorEstimation {t1-t0:.3f} s ({' '.join(l.split()[1]+'='+l.split()[2] for l in r.stderr.splitlines() if l.startswith('TIMING'))}), AmpliSolveVariantCalling {t2-t1:.3f} s")
print(sum(1 for _ in open(f"{O}/vc/Summary_Variant_Info.txt")) - 1, "calls")
