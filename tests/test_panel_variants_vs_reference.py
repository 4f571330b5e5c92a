"""Inputs the reference accepts in more than one shape, through the reference's own code and through ours on the CPU: the BED rows in
any order, with carriage returns, one amplicon listed twice; ASEQ lines in another order than the panel's, lines of positions the
panel does not have, a position listed three times, a file that is only a header, a sample without a whole amplicon, a chromosome name with underscores in it.  Reference =
oracle/_ref/ee_ref_driver (storeGermlineStatistics + estimateThresholds + generateFinalOutput compiled where the sources lie, EE:1057-2944);
ours = the C++ host's panel walk + reader (csrc/host/panel.cpp, aseq.cpp) -> the CPU oracle standing in for the kernels -> the C++
table writer (csrc/host/table.cpp).  The tables must be the same bytes.  (The kernels against the oracle: the -m gpu tests.)"""
import os
import subprocess

import numpy as np
import pytest

from amplisolve_amd.hostio import HostCohort
from oracle import pyoracle as orc
from tests.helpers import write_fresh_panel

pytestmark = pytest.mark.skipif(not os.path.exists(orc.REF_EE_DRIVER), reason="oracle/_ref/ee_ref_driver is absent (make -C oracle where /root/reference exists)")
HEADER = "chr\tpos\tdbsnp\tMAF\tref\talt\tA\tC\tG\tT\tRD\tArs\tCrs\tGrs\tTrs\n"


def _vary(d, rng, what, sub="N"):
    """sub: the directory of ASEQ files the aseq_* shapes are applied to (the bed_* shapes rewrite the panel and its by-products)"""
    bed = (d / "p.bed").read_text().splitlines()
    files = sorted(os.listdir(d / sub))
    if "bed_shuffled" in what:
        rng.shuffle(bed)
    if "bed_twice" in what:
        bed.append(bed[0])  # the first amplicon once more: every one of its positions is then a duplicated row of the table
    (d / "p.bed").write_text("".join(r + ("\r\n" if "bed_crlf" in what else "\n") for r in bed))
    if "bed_twice" in what or "bed_shuffled" in what:
        # the by-products of the reference's own panel step (EE:578-670), which its driver takes as files: the walk's reference bases and
        # the positions the walk visits more than once
        base = dict((tuple(l.split()[:2]), l.split()[2]) for l in (d / "r.txt").read_text().splitlines())
        walk = [(c, str(p)) for c, a, b in (r.split("\t")[:3] for r in bed) for p in range(int(a), int(b) + 1)]
        (d / "r.txt").write_text("".join(f"{c}\t{p}\t{base[(c, p)]}\n" for c, p in walk))
        seen, dups = set(), []
        for k in walk:
            if k in seen and k not in dups:
                dups.append(k)
            seen.add(k)
        (d / "d.txt").write_text("".join(f"{c}\t{p}\n" for c, p in sorted(dups, key=lambda k: (k[0], int(k[1])))))
    for k, f in enumerate(files):
        p = d / sub / f
        lines = p.read_text().splitlines()[1:]
        if "aseq_shuffled" in what and k % 2 == 0:
            rng.shuffle(lines)
        if "aseq_offpanel" in what:
            extra = [f"chr9\t{5000 + j}\t.\t.\t.\t.\t500\t1\t0\t2\t503\t250\t0\t0\t1" for j in range(7)] + \
                    [l.replace("\t", "\t9", 1) for l in lines[:5]]  # a coordinate far off the amplicons, on a chromosome of the panel
            for e in extra:
                lines.insert(int(rng.integers(0, len(lines) + 1)), e)
        if "aseq_triple" in what and k == 1:
            tok = lines[3].split("\t")
            tok[6], tok[10] = str(int(tok[6]) + 1000), str(int(tok[10]) + 1000)  # 1000 more forward A reads (total and RD columns): a well-formed other line
            lines += [lines[3], "\t".join(tok), lines[10]]  # lines 3 and 10 again: listed three times / twice
        if "aseq_own_rd" in what:  # lines whose RD column is not A+C+G+T (EE:1178-1181: used with their own RD, EE:1229), some of them 0
            for i in range(len(lines)):
                if rng.random() < 0.04:
                    tok = lines[i].split("\t")
                    tok[10] = str(0 if rng.random() < 0.15 else max(0, int(tok[10]) + int(rng.integers(-40, 400))))
                    lines[i] = "\t".join(tok)
        if "aseq_header_only" in what and k == 2:
            lines = []
        if "aseq_no_amplicon" in what and k == 3:
            first = bed[0].split("\t")[0:3]
            lines = [l for l in lines if not (l.split("\t")[0] == first[0] and int(first[1]) <= int(l.split("\t")[1]) <= int(first[2]))]
        p.write_text(HEADER + "".join(l + "\n" for l in lines))


def _rename_chromosome(d):
    """a chromosome name with underscores: the reference cuts its keys at underscores (EE:1552 `%[^_]_%[^_]_%[^_]`)"""
    for f in [d / "p.bed", d / "r.txt", d / "d.txt"] + [d / sub / x for sub in ("N", "T") if (d / sub).is_dir() for x in os.listdir(d / sub)]:
        f.write_text(f.read_text().replace("chr7\t", "chr7_KI270803v1_alt\t"))


@pytest.mark.parametrize("seed,what", [(21, ("bed_shuffled",)), (27, ("chrom_underscores",)), (22, ("bed_crlf", "aseq_offpanel")), (23, ("bed_twice",)), (24, ("aseq_shuffled", "aseq_triple")),
                                       (25, ("aseq_header_only", "aseq_no_amplicon", "aseq_offpanel")), (26, ("bed_shuffled", "bed_twice", "aseq_shuffled", "aseq_triple", "aseq_header_only")), (28, ("aseq_own_rd",)), (29, ("aseq_own_rd", "aseq_triple", "bed_twice"))])
def test_table_equals_the_references_on_input_variants(tmp_path, monkeypatch, seed, what):
    rng = np.random.default_rng(seed)
    d = tmp_path
    write_fresh_panel(d, seed, S=7, amplicons=5)
    if "chrom_underscores" in what:
        _rename_chromosome(d)
    _vary(d, rng, what)
    (d / "o").mkdir()
    r = subprocess.run([orc.REF_EE_DRIVER, "p.bed", "r.txt", "d.txt", "N", "0.002", "100", "o", "dump"], capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stdout[-400:] + r.stderr[-400:]
    name = [n for n in os.listdir(d / "o") if n.startswith("positionSpecificNoise_")]
    assert len(name) == 1
    want = (d / "o" / name[0]).read_bytes()
    monkeypatch.chdir(d)  # the same directory literal as the reference's run: the sample visit order hangs on it (a1)
    co = HostCohort("p.bed", "N", refbases_file="r.txt")
    assert co.names == (d / "dump.order").read_text().split()  # the reference's own visit order of these files (EE:1081)
    acc = orc.error_reduce(co.recs, co.P, 0.002, 100, E=co.E, dup_off=co.dup_off, rd=co.rd_plane())
    fin = orc.error_finalize(acc)
    co.write_error_table(fin["rate"], fin["code"], fin["germ_val"].astype(np.float32), fin["germ_present"], "ours.txt")
    got = (d / "ours.txt").read_bytes()
    if "aseq_own_rd" in what:
        assert co.stats()["irregular"] > 100
    assert len(want) > 10_000
    assert got == want


@pytest.mark.skipif(not os.path.exists(getattr(orc, "REF_VC_DRIVER", "")), reason="oracle/_ref/vc_ref_driver is absent")
@pytest.mark.parametrize("seed,what", [(31, ("bed_twice", "aseq_triple")), (32, ("bed_shuffled", "chrom_underscores")), (33, ("bed_crlf", "aseq_offpanel", "aseq_header_only"))])
def test_calling_side_reads_those_tables_as_the_reference_does(tmp_path, seed, what):
    """The calling program's view of the table the reference writes for such a panel: our reader's reference cell, duplicate flag,
    threshold and germ-max cells == the four maps the reference's own storeInputFile fills (VC:430-576), its by-product VCF byte for
    byte, and the 10-mers / homopolymer flag of every position x 4 substituted bases == the reference's find_kmer_down / find_kmer_up /
    homopolymerTest (VC:3307-3718) -- live, on fresh tables (tests/test_oracle_golden.py does the same on ten committed ones)."""
    from amplisolve_amd.hostio import ErrorTable

    rng = np.random.default_rng(seed)
    d = tmp_path
    write_fresh_panel(d, seed, S=5, amplicons=4)
    if "chrom_underscores" in what:
        _rename_chromosome(d)
    _vary(d, rng, what)
    (d / "o").mkdir()
    r = subprocess.run([orc.REF_EE_DRIVER, "p.bed", "r.txt", "d.txt", "N", "0.002", "100", "o"], capture_output=True, text=True, cwd=d)
    assert r.returncode == 0
    table = "o/" + [n for n in os.listdir(d / "o") if n.startswith("positionSpecificNoise_")][0]
    maps = subprocess.run([orc.REF_VC_DRIVER, "maps", table, "ref_dummy.vcf"], capture_output=True, cwd=d)
    ctx = subprocess.run([orc.REF_VC_DRIVER, "context", table, "ref_dummy2.vcf"], capture_output=True, cwd=d)
    assert maps.returncode == 0 and ctx.returncode == 0
    t = ErrorTable(str(d / table), dummy_vcf=str(d / "our_dummy.vcf"))
    lines = {"R": [], "D": [], "T": [], "G": []}
    for p in range(t.P):
        c, x = t.key(p)
        key = f"{c}_{x}"
        lines["R"].append(f"{key} {t.cell(p, 0)}")
        if t.dup[p]:
            lines["D"].append(f"{key} {x}")
        for nt in range(4):
            lines["T"].append(f"{key}_{'ACGT'[nt]} {t.cell(p, 1 + nt)}")
            lines["G"].append(f"{key}_{'ACGT'[nt]} {t.cell(p, 5 + nt)}")
    ours = "".join(f"{tag} {l}\n" for tag in "RDTG" for l in sorted(lines[tag])).encode()
    assert ours == maps.stdout and len(ours.splitlines()) > 5000
    assert (d / "our_dummy.vcf").read_bytes() == (d / "ref_dummy.vcf").read_bytes()
    want = ctx.stdout.decode().splitlines()
    assert len(want) == t.P
    for p, w in enumerate(want):
        f = w.split(" ")
        c, x = t.key(p)
        assert (f[0], f[1], f[2]) == ("C", c, str(x))
        for i, sub in enumerate("ACGT"):
            down, up, flag = t.context(p, sub)
            assert (down, up, str(flag)) == (f[3], f[4], f[5 + i]), (w, sub)


@pytest.mark.skipif(not os.path.exists(getattr(orc, "REF_VC_DRIVER", "")), reason="oracle/_ref/vc_ref_driver is absent")
@pytest.mark.parametrize("n,seed", [(1, 1), (7, 2), (13, 3), (60, 4), (200, 5)])
def test_visit_order_of_freshly_named_files(tmp_path, monkeypatch, n, seed):
    """a1 live: the order the files of a directory are visited in is the iteration order of the reference's own unordered_map of
    {listed path -> sample name} (VC:387-394, 580-627, 672; EE twin EE:552-559, 794-841) -- for file names of any length and for
    counts on both sides of the map's rehash points, not only for the committed fixtures."""
    from amplisolve_amd.hostio import sample_order

    rng = np.random.default_rng(seed)
    d = tmp_path / "T"
    d.mkdir()
    alphabet = list("ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789-.")
    names = set()
    while len(names) < n:
        names.add("".join(rng.choice(alphabet, int(rng.integers(1, 24)))).lstrip(".-") or "x")
    for nm in names:
        (d / f"{nm}.PILEUP.ASEQ").write_text("chr\tpos\n")
    (d / "notes.txt").write_text("not an ASEQ file\n")
    monkeypatch.chdir(tmp_path)
    r = subprocess.run([orc.REF_VC_DRIVER, "order", "T", "list.txt"], capture_output=True, text=True, cwd=tmp_path)
    assert r.returncode == 0
    want = [l.split("\t")[0] for l in r.stdout.splitlines()]
    assert sorted(want) == sorted(names)
    assert sample_order("T") == want


@pytest.mark.parametrize("seed,what,err", [(41, ("bed_shuffled",), "0.01"), (42, ("bed_twice", "bed_crlf"), "0.0375"), (43, ("chrom_underscores",), "1e-3"),
                                           (44, ("bed_shuffled", "bed_twice"), "-1"), (45, (), "0.5x"), (46, ("bed_crlf",), "2")])
def test_default_table_of_such_panels_through_the_command_line(tmp_path, seed, what, err):
    """germline_dir=not_available (EE:472-506, generateFinalOutput_default EE:2948-3043; no GPU in this mode): the drop-in executable
    beside the reference's own code on fresh panels in those shapes and default_error strings as main() converts them (atof; <= 0 -> 0.01)."""
    rng = np.random.default_rng(seed)
    d = tmp_path
    write_fresh_panel(d, seed, S=2, amplicons=6)
    if "chrom_underscores" in what:
        _rename_chromosome(d)
    _vary(d, rng, what)
    (d / "o").mkdir()
    import ctypes as C

    libc = C.CDLL(None)
    libc.atof.restype = C.c_double
    v = libc.atof(err.encode())
    conv = repr(float(np.float32(v))) if v > 0 else "0.01"  # what main() hands on (EE:353-363): atof into a float, 0.01 for a value <= 0
    r = subprocess.run([orc.REF_EE_DRIVER, "--default", "p.bed", "r.txt", "d.txt", conv, "o"], capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stdout[-300:] + r.stderr[-300:]
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "amplisolve_amd", "bin", "AmpliSolveErrorEstimation")
    q = subprocess.run([exe, "panel_design=p.bed", "reference_genome=unused.fa", "germline_dir=not_available", "C_value=0.002", "coverage_cutoff=100",
                        f"default_error={err}", "output_dir=q"], capture_output=True, text=True, cwd=d,
                       env=dict(os.environ, AMPLISOLVE_STRICT_EXIT="1", AMPLISOLVE_REFBASES_FILE="r.txt"))
    assert q.returncode == 0, q.stdout[-300:] + q.stderr[-300:]
    want, got = (d / "o" / "positionSpecificNoise_default.txt").read_bytes(), (d / "q" / "positionSpecificNoise_default.txt").read_bytes()
    assert len(want) > 5000 and got == want
