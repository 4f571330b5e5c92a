"""The reference's own callVariants (VC:633-3304), compiled where it lies without its Fisher statements (oracle/Makefile VC_CALL_DROP:
the Boost include, fisherTest and the 12 `p=fisherTest(...)` lines are left out, nothing is rewritten, nothing stands in for Boost;
p keeps the -1 of VC:901 so FisherPvalue prints -1 and the Fisher flag is YES), against
  * the survey's digest of the reference's run on Toy_data (the build really is the reference's callVariants outside columns 13-14),
  * the CPU oracle's restated per-line gate (a6) on freshly drawn panels -- CPU, every round,
  * the drop-in command line on the GPU: Summary_Variant_Info.txt and every <sample>.vcf byte for byte with AMPLISOLVE_FISHER=off,
    and outside the Fisher-dependent columns as shipped (tools/fuzz_cli_vc_vs_reference.py is the same comparison at length)."""
import hashlib
import os
import subprocess

import numpy as np
import pytest

from amplisolve_amd.hostio import HostCohort, read_error_table
from oracle import pyoracle as orc
from tests.helpers import write_fresh_panel, write_fresh_tumours
from tests.test_panel_variants_vs_reference import _rename_chromosome, _vary

G = "/root/repo/tests/golden"
need_ref = pytest.mark.skipif(not (os.path.exists(orc.REF_VC_NOFISHER) and os.path.exists(orc.REF_EE_DRIVER)),
                              reason="oracle/_ref/AmpliSolveVariantCalling_noFisher is absent (make -C oracle where /root/reference exists)")


@need_ref
@pytest.mark.skipif(not os.path.isdir("/root/reference/Toy_data"), reason="full Toy_data only exists in the build container")
def test_reference_callvariants_without_fisher_reproduces_the_survey_digest(tmp_path):
    """sha256 of the reference's Summary on Toy_data with columns 13-14 cut away, taken from a run of the whole
    program.  The build without the Fisher statements must give the same digest: what was left out touches nothing else."""
    import gzip

    (tmp_path / "psn.txt").write_bytes(gzip.open(f"{G}/toy/positionSpecificNoise_0.0020.txt.gz").read())
    r = subprocess.run([orc.REF_VC_NOFISHER, "errorFile=psn.txt", "tumour_dir=/root/reference/Toy_data/TUMOUR_ASEQ_DIR", "output_dir=o",
                        "coverage_cutoff=100", "p_value=0.05"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0, r.stdout[-500:]
    lines = (tmp_path / "o" / "Summary_Variant_Info.txt").read_text().splitlines()
    assert len(lines) == 711 and all(l.split("\t")[13] == "-1" and l.split("\t")[12].endswith("_YES") for l in lines[1:])
    cut = "".join("\t".join(l.split("\t")[:12] + l.split("\t")[14:]) + "\n" for l in lines)
    assert hashlib.sha256(cut.encode()).hexdigest() == "4a324a65984ab3b296507a2dab7687b89d298aa87813d8e6edaab0d7deb5c13e"
    # the same call sequence with a clock around each step (bench.py's cpu_baseline) writes the same file
    (tmp_path / "o2" / "AmpliSolveVariantCalling_interm_files").mkdir(parents=True)
    t = subprocess.run([orc.REF_VC_CALL_DRIVER, "time", "psn.txt", "/root/reference/Toy_data/TUMOUR_ASEQ_DIR", "o2", "100", "0.05"], capture_output=True, text=True, cwd=tmp_path)
    assert t.returncode == 0 and "TIMING callVariants" in t.stderr
    assert (tmp_path / "o2" / "Summary_Variant_Info.txt").read_bytes() == (tmp_path / "o" / "Summary_Variant_Info.txt").read_bytes()


def _fresh(d, seed, what, S=6, T=3, depth=2000, amplicons=5):
    rng = np.random.default_rng(seed)
    write_fresh_panel(d, seed, depth=depth, S=S, amplicons=amplicons)
    write_fresh_tumours(d, seed, T=T, depth=depth)
    if "chrom_underscores" in what:
        _rename_chromosome(d)
    _vary(d, rng, what, sub="N")
    _vary(d, rng, tuple(x for x in what if x.startswith("aseq_")), sub="T")
    if "ref_N" in what or "ref_soft" in what:
        from tools.fuzz_cli_vc_vs_reference import soften_reference

        soften_reference(d, rng, what)
    (d / "o").mkdir()
    r = subprocess.run([orc.REF_EE_DRIVER, "p.bed", "r.txt", "d.txt", "N", "0.002", "100", "o"], capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stderr[-500:]
    return "o/" + [n for n in os.listdir(d / "o") if n.startswith("positionSpecificNoise_")][0]


@need_ref
@pytest.mark.parametrize("seed,cov,what", [(41, "100", ()), (42, "30", ("bed_twice", "aseq_triple")), (43, "100", ("aseq_own_rd",)), (44, "400", ("aseq_shuffled", "aseq_offpanel", "aseq_header_only")),
                                           (45, "100", ("ref_N", "ref_soft")), (46, "1", ("chrom_underscores", "bed_crlf", "aseq_own_rd", "aseq_triple"))])
def test_oracle_gate_equals_the_references_callvariants(tmp_path, monkeypatch, seed, cov, what):
    """a6 pinned: the rows the reference's callVariants emits for fresh tumour files (sample, position, substitution, RD / FW / BW,
    the three VAFs and both Q as printed, in the reference's order) == what the oracle's restated gate (oracle_poisson_call_rd,
    VC:752-898) computes from our host's packed records and the table as our reader loads it."""
    d = tmp_path
    table = _fresh(d, seed, what)
    r = subprocess.run([orc.REF_VC_NOFISHER, f"errorFile={table}", "tumour_dir=T", "output_dir=rv", f"coverage_cutoff={cov}", "p_value=0.05"], capture_output=True, text=True, cwd=d)
    assert r.returncode == 0
    body = [l.split("\t") for l in (d / "rv" / "Summary_Variant_Info.txt").read_text().splitlines()[1:]]
    monkeypatch.chdir(d)  # same directory literal -> same visit order of the tumour files (VC:672)
    co = HostCohort(table, "T", is_error_table=True, keep_line_no=True)
    ref, thr = read_error_table(table)
    exp = orc.poisson_call(co.recs, co.P, thr, ref, int(cov), E=co.E, ext_pos=co.ext_pos, rd=co.rd_plane())
    rdp = co.rd_plane()
    want = []
    for t, name in enumerate(co.names):
        for _, r_ in sorted((co.line_no[t, r_], r_) for r_ in range(co.P + co.E) if exp["call_mask"][t, r_]):
            p = r_ if r_ < co.P else co.ext_pos[r_ - co.P]
            c, x = co.position(p)
            rec = co.recs[t, r_].astype(np.int64)
            own = rdp is not None and rdp[t, r_] != np.iinfo(np.int32).min
            RD = int(rdp[t, r_]) if own else int(rec.sum())
            for a in range(4):
                if exp["call_mask"][t, r_] >> a & 1:
                    want.append([name, c, str(x), f"{'ACGT'[ref[p]]}->{'ACGT'[a]}", str(RD), str(int(rec[:4].sum())), str(int(rec[4:].sum())),
                                 exp["af"][t, r_, a], str(int(rec[a])), str(int(rec[4 + a])), exp["q"][t, r_, a]])
    assert len(body) == len(want) and len(want) > 20
    for i, (g, w) in enumerate(zip(body, want)):
        prec = 6 if i == 0 else 4  # VC:1066: the precision set inside the first row sticks
        assert g[:7] == w[:7] and g[8:10] == w[8:10], (g, w)
        assert [g[7], g[10], g[11]] == [f"{float(v):.{prec}g}" for v in w[7]], (g, w)
        assert [g[14], g[15]] == [f"{q:.4g}" for q in w[10]], (g, w)
    if "aseq_own_rd" in what:
        assert sum(1 for g in body if int(g[4]) != int(g[5]) + int(g[6])) > 0  # rows whose RD column is the line's own


@need_ref
@pytest.mark.gpu
@pytest.mark.parametrize("seed", [7001, 7002, 7003, 7004, 7005, 7006, 7007, 7008, 7009, 7010, 7011, 7012])
def test_command_line_equals_the_references_callvariants(seed):
    """Both command lines on a fresh panel beside the reference's own code (shapes, cut-offs, chunk sizes, threads and record layouts
    drawn from the seed): error table identical, then Summary + VCFs byte for byte with the Fisher statement left out on both sides,
    and outside columns 13-14 / FILTER with our Fisher in."""
    from tools.fuzz_cli_vc_vs_reference import one_case

    ok, line, rows = one_case(seed)
    assert ok, line
