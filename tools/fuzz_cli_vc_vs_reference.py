"""Freshly drawn panels + tumour files through the REFERENCE's own variant caller and through the drop-in command line, in the same
directory with the same literals.  Reference = oracle/_ref/AmpliSolveVariantCalling_noFisher: the reference's whole translation unit,
main() included, compiled where it lies minus its Boost include, fisherTest and the 12 `p=fisherTest(...)` statements ("callVariants
without its Fisher statements", oracle/Makefile VC_CALL_DROP; p stays -1, the Fisher flag YES).  Two comparisons per case:
  (a) ours with AMPLISOLVE_FISHER=off (p stays -1 too): Summary_Variant_Info.txt and every <sample>.vcf must be the SAME BYTES
      (but the ##fileDate line) -- gate, VAFs, sticky precision, flag order out of the unordered_map, C->G "-" ID, file order;
  (b) ours as shipped (own Fisher): the Summary but columns 13-14 and the VCFs but the FILTER and ID columns must be the same.
The error table of each case comes from the reference's error estimation (ee_ref_driver) and from ours, which must agree first.
Shapes drawn per case: positions listed twice (overlapping amplicons, always), a BED amplicon listed twice, lines with their own RD
column, positions listed three times, shuffled lines, off-panel lines, a header-only file, N / soft-masked reference bases, a
chromosome name with underscores, CRLF BED, coverage cut-offs 1..2000, p-values, chunk sizes, parser threads, record layouts.
Usage (GPU box; /root/reference not needed -- the reference builds are prebuilt files): python tools/fuzz_cli_vc_vs_reference.py [cases] [first seed]"""
import os
import pathlib
import subprocess
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import pyoracle as orc
from tests.helpers import write_fresh_panel, write_fresh_tumours
from tests.test_panel_variants_vs_reference import _rename_chromosome, _vary

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "amplisolve_amd", "bin")


def strip_date(b):
    return b"\n".join(l for l in b.split(b"\n") if not l.startswith(b"##fileDate="))


def cut_fisher(b):  # Summary without columns 13-14 (AmpliconEdge_StrandBias, FisherPvalue)
    return b"\n".join(b"\t".join(l.split(b"\t")[:12] + l.split(b"\t")[14:]) for l in b.split(b"\n"))


def cut_filter(b):
    """VCF body without the FILTER column (its StrandBias / PositionWithHighNoise tokens hang on Fisher) and without the ID column (the
    C->G block writes "-" instead of "." when the row is not a PASS, VC:1856 -- and in the build without Fisher every row carries StrandBias)"""
    return b"\n".join(l if l.startswith(b"#") else b"\t".join(l.split(b"\t")[:2] + l.split(b"\t")[3:6] + l.split(b"\t")[7:]) for l in strip_date(b).split(b"\n"))


def soften_reference(d, rng, what):
    """N / lower-case reference bases (both are compared case-sensitively with "A".."T", EE:2668-2670, VC:869: no -2_-2 cell, no call)"""
    rows = [l.split("\t") for l in (d / "r.txt").read_text().splitlines()]
    keys = sorted({(r[0], r[1]) for r in rows})
    pick = {k: ("N" if rng.random() < 0.5 else None) for k in keys if rng.random() < 0.06}
    for r in rows:
        k = (r[0], r[1])
        if k in pick:
            r[2] = "N" if ("ref_N" in what and pick[k]) else (r[2].lower() if "ref_soft" in what else r[2])
    (d / "r.txt").write_text("".join("\t".join(r) + "\n" for r in rows))


def one_case(seed, keep=None):
    rng = np.random.default_rng(seed)
    depth = int(rng.choice([300, 2000, 6000, 30_000, 120_000]))
    S, T = int(rng.integers(3, 12)), int(rng.integers(1, 7))
    C = f"{float(rng.choice([0.0005, 0.002, 0.01])):g}"
    cov = str(int(rng.choice([1, 30, 100, 400, 2000])))
    pv = str(rng.choice(["0.05", "0.01", "0.5", "7"]))  # 7 -> converted to the default (VC:286-290)
    shapes_all = ["bed_twice", "bed_shuffled", "bed_crlf", "aseq_own_rd", "aseq_triple", "aseq_shuffled", "aseq_offpanel", "aseq_header_only",
                  "aseq_no_amplicon", "ref_N", "ref_soft", "chrom_underscores"]
    what = tuple(x for x in shapes_all if rng.random() < 0.3)
    env = dict(os.environ, AMPLISOLVE_STRICT_EXIT="1", AMPLISOLVE_REFBASES_FILE="r.txt", AMPLISOLVE_TIMING="1", AMPLISOLVE_THREADS=str(int(rng.integers(1, 9))))
    if rng.random() < 0.7:
        env["AMPLISOLVE_CHUNK_BYTES"] = str(int(rng.choice([1, 40_000, 200_000, 2_000_000])))
    if rng.random() < 0.4:
        env["AMPLISOLVE_RECORDS"] = str(rng.choice(["u16", "u24", "i32"]))
    with tempfile.TemporaryDirectory(prefix="ampli_vcfuzz_") as td:
        d = pathlib.Path(td)
        write_fresh_panel(d, seed, depth=depth, S=S, amplicons=int(rng.integers(3, 9)))
        lines = write_fresh_tumours(d, seed, T=T, depth=int(depth * float(rng.choice([0.5, 1, 3]))))
        if "chrom_underscores" in what:
            _rename_chromosome(d)
        _vary(d, rng, what, sub="N")
        _vary(d, rng, tuple(x for x in what if x.startswith("aseq_")), sub="T")
        if "ref_N" in what or "ref_soft" in what:
            soften_reference(d, rng, what)
        (d / "o").mkdir()
        ee_ref = subprocess.run([orc.REF_EE_DRIVER, "p.bed", "r.txt", "d.txt", "N", C, cov, "o"], capture_output=True, text=True, cwd=d)
        ee_ours = subprocess.run([f"{BIN}/AmpliSolveErrorEstimation", "panel_design=p.bed", "reference_genome=x.fa", "germline_dir=N", f"C_value={C}",
                                  f"coverage_cutoff={cov}", "default_error=0.01", "output_dir=q"], capture_output=True, text=True, cwd=d, env=env)
        names = [n for n in os.listdir(d / "o") if n.startswith("positionSpecificNoise_")] if ee_ref.returncode == 0 else []
        msg = []
        if not (ee_ours.returncode == 0 and len(names) == 1 and (d / "o" / names[0]).read_bytes() == (d / "q" / names[0]).read_bytes()):
            return False, f"seed {seed}: error tables differ (reference rc {ee_ref.returncode}, ours rc {ee_ours.returncode}) {what}", 0
        table = f"o/{names[0]}"
        ref = subprocess.run([orc.REF_VC_NOFISHER, f"errorFile={table}", "tumour_dir=T", "output_dir=rv", f"coverage_cutoff={cov}", f"p_value={pv}"],
                             capture_output=True, text=True, cwd=d)
        args = [f"{BIN}/AmpliSolveVariantCalling", f"errorFile=q/{names[0]}", "tumour_dir=T", None, f"coverage_cutoff={cov}", f"p_value={pv}"]
        off = subprocess.run(args[:3] + ["output_dir=v0"] + args[4:], capture_output=True, text=True, cwd=d, env=dict(env, AMPLISOLVE_FISHER="off"))
        on = subprocess.run(args[:3] + ["output_dir=v1"] + args[4:], capture_output=True, text=True, cwd=d, env=env)
        ok = ref.returncode == 0 and off.returncode == 0 and on.returncode == 0
        rows = 0
        if ok:
            want = sorted(os.listdir(d / "rv"))
            ok = want == sorted(os.listdir(d / "v0")) == sorted(os.listdir(d / "v1"))
            if not ok:
                msg.append(f"file lists differ: {want} / {sorted(os.listdir(d / 'v0'))}")
            for n in want if ok else []:
                if n == "Summary_Variant_Info.txt":
                    a, b0, b1 = [(d / x / n).read_bytes() for x in ("rv", "v0", "v1")]
                    rows = a.count(b"\n") - 1
                    if a != b0:
                        msg.append("Summary differs (Fisher off)")
                    if cut_fisher(a) != cut_fisher(b1):
                        msg.append("Summary differs outside columns 13-14 (own Fisher)")
                elif n.endswith(".vcf"):
                    a, b0, b1 = [(d / x / n).read_bytes() for x in ("rv", "v0", "v1")]
                    if strip_date(a) != strip_date(b0):
                        msg.append(f"{n} differs (Fisher off)")
                    if cut_filter(a) != cut_filter(b1):
                        msg.append(f"{n} differs outside FILTER (own Fisher)")
                elif n == "AmpliSolveVariantCalling_interm_files":
                    a, b0 = [(d / x / n / "dummyVCF_1.vcf").read_bytes() for x in ("rv", "v0")]
                    if a != b0:
                        msg.append("dummyVCF_1.vcf differs")
            ok = not msg
        else:
            why = [ln.strip() for ln in (off.stdout + on.stdout).splitlines() if "went wrong" in ln.lower() or "error" in ln.lower()]
            msg.append(f"rc reference {ref.returncode} ours {off.returncode}/{on.returncode}: " + ("; ".join(why)[:600] or (off.stdout + off.stderr)[-400:]))
        if not ok and keep:
            subprocess.run(["cp", "-r", str(d), keep])
        line = (f"seed {seed}: depth {depth:>7} S {S:2d} T {T} lines {lines:5d} C {C:>6} cov {cov:>4} p {pv:>4} threads {env['AMPLISOLVE_THREADS']} "
                f"chunk_bytes {env.get('AMPLISOLVE_CHUNK_BYTES', 'default'):>8} records {env.get('AMPLISOLVE_RECORDS', 'auto'):>4} calls {rows:5d}  "
                f"{'IDENTICAL' if ok else 'NOT IDENTICAL'} {','.join(what)}")
        return ok, line + ("".join("\n   " + m for m in msg)), rows


if __name__ == "__main__":
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 6000
    for need in (orc.REF_EE_DRIVER, orc.REF_VC_NOFISHER):
        assert os.path.exists(need), f"{need} is absent (make -C oracle where /root/reference exists)"
    bad = total = 0
    for case in range(n_cases):
        ok, line, rows = one_case(seed0 + case, keep=os.environ.get("FUZZ_KEEP"))
        print(line, flush=True)
        bad += 0 if ok else 1
        total += rows
    print(f"{n_cases} panels, {total} calls compared, {bad} different")
    sys.exit(1 if bad else 0)
