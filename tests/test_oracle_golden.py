"""The CPU oracle (and the C++ host's parsers / writers) against the reference's own outputs.

Fixtures under tests/golden/ were produced by tests/golden/make_golden.py from the compiled reference
(oracle/_ref/ee_ref_driver, oracle/_ref/libvc_scorer_ref.so).  No GPU involved.
"""
import gzip
import hashlib
import os
import tempfile

import numpy as np
import pytest

from amplisolve_amd.hostio import HostCohort, read_error_table, sample_order
from oracle import pyoracle as orc

G = "/root/repo/tests/golden"  # literal: the sample visit order hashes the directory string
pytestmark = pytest.mark.skipif(not os.path.isdir(G), reason="fixtures are addressed through the /root/repo path")


def oracle_table(co, C, cov, tmp_path, name="t.txt"):
    acc = orc.error_reduce(co.recs, co.P, C, cov, E=co.E, dup_off=co.dup_off)
    assert acc["order_sensitive"] == 0
    fin = orc.error_finalize(acc)
    out = str(tmp_path / name)
    co.write_error_table(fin["rate"], fin["code"], fin["germ_val"].astype(np.float32), fin["germ_present"], out)
    return acc, fin, open(out).read()


@pytest.mark.parametrize("C,cov,expected", [(0.002, 100, "expected_positionSpecificNoise_0.0020.txt"),
                                            (0.01, 500, "expected_positionSpecificNoise_0.0100_cov500.txt")])
def test_toy_subset_table_is_byte_identical(tmp_path, C, cov, expected):
    d = f"{G}/toy_subset"
    co = HostCohort(f"{d}/panel.bed", f"{d}/NORMAL", refbases_file=f"{d}/refbases.txt")
    assert co.names == open(f"{d}/expected_visit_order.txt").read().split()
    assert co.E > 0 and co.stats()["offpanel"] == 0
    _, _, got = oracle_table(co, C, cov, tmp_path)
    assert got == open(f"{d}/{expected}").read()


def test_toy_subset_integer_counts_match_reference(tmp_path):
    """Count_Hash of the reference (the integer quorum counts, EE:1665) == oracle cnt, key by key."""
    d = f"{G}/toy_subset"
    co = HostCohort(f"{d}/panel.bed", f"{d}/NORMAL", refbases_file=f"{d}/refbases.txt")
    acc = orc.error_reduce(co.recs, co.P, 0.002, 100, E=co.E, dup_off=co.dup_off)
    exp = dict(l.split() for l in open(f"{d}/expected_counts.txt"))
    assert len(exp) == 4 * co.P
    for p in range(co.P):
        c, x = co.position(p)
        for nt in range(4):
            assert int(exp[f"{c}_{x}_{'ACGT'[nt]}"]) == acc["cnt"][nt, p]


@pytest.mark.parametrize("tag,C,cov", [("0.0020_cov100", 0.002, 100), ("0.0005_cov1", 0.0005, 1), ("0.0500_cov1000", 0.05, 1000)])
def test_mini_edge_table_is_byte_identical(tmp_path, tag, C, cov):
    """Synthetic edge cases: absent lines, duplicated positions, FW or BW = 0 (NaN gate), depth around the cutoff,
    AF around 5 %, quorum edges, chrX / chrM, N and soft-masked reference bases."""
    d = f"{G}/mini_edge"
    co = HostCohort(f"{d}/panel.bed", f"{d}/NORMAL", refbases_file=f"{d}/refbases.txt")
    assert co.names == open(f"{d}/expected_visit_order.txt").read().split()
    acc, fin, got = oracle_table(co, C, cov, tmp_path)
    assert got == open(f"{d}/expected_positionSpecificNoise_{tag}.txt").read()
    exp = dict(l.split() for l in open(f"{d}/expected_counts_{tag}.txt"))
    for p in range(co.P):
        c, x = co.position(p)
        for nt in range(4):
            assert int(exp[f"{c}_{x}_{'ACGT'[nt]}"]) == acc["cnt"][nt, p]
    if cov <= 100:
        assert set(np.unique(fin["code"])) >= {0, 1}


VC_TABLES = {  # name under tests/golden/vc_ref/ -> the table the reference's storeInputFile was run on (make_golden_vc.py)
    "toy_subset_0.0020": "toy_subset/expected_positionSpecificNoise_0.0020.txt",
    "toy_subset_0.0100_cov500": "toy_subset/expected_positionSpecificNoise_0.0100_cov500.txt",
    "toy_subset_default_0.0120": "toy_subset/expected_positionSpecificNoise_default_0.0120.txt",
    "mini_edge_0.0020_cov100": "mini_edge/expected_positionSpecificNoise_0.0020_cov100.txt",
    "mini_edge_0.0005_cov1": "mini_edge/expected_positionSpecificNoise_0.0005_cov1.txt",
    "mini_edge_0.0500_cov1000": "mini_edge/expected_positionSpecificNoise_0.0500_cov1000.txt",
    "mini_edge_default_7": "mini_edge/expected_positionSpecificNoise_default_7.txt",
    "irregular_0.0020_cov100": "irregular/expected_positionSpecificNoise_0.0020_cov100.txt",
    "context_edge": "vc_ref/context_edge.txt",
    "toy_full_0.0020": "toy/positionSpecificNoise_0.0020.txt.gz",
}


def _vc_digests():
    out = {}
    for l in open(f"{G}/vc_ref/digests.txt"):
        name, _, sha, _, n = l.split()
        out[name] = (sha, int(n))
    return out


def _vc_table(name, tmp_path):
    src = f"{G}/{VC_TABLES[name]}"
    if not src.endswith(".gz"):
        return src
    dst = tmp_path / f"{name}.txt"
    dst.write_bytes(gzip.open(src).read())
    return str(dst)


def _strtof(text):
    """glibc strtof -- what std::stof (VC:889-890) calls -- as the float's bit pattern"""
    import ctypes as C

    libc = C.CDLL(None)
    libc.strtof.restype = C.c_float
    libc.strtof.argtypes = [C.c_char_p, C.c_void_p]
    return np.float32(libc.strtof(text.encode(), None)).view(np.int32)


@pytest.mark.parametrize("name", list(VC_TABLES))
def test_error_table_reader_against_the_reference_maps(tmp_path, name):
    """a5, reference-pinned: our reader's view of a table == the four maps the reference's own storeInputFile (VC:430-576,
    compiled from the reference, oracle/_ref/vc_ref_driver) fills from the same file -- reference cell, duplicate entry,
    threshold cell and germ-max cell of every key, the first row of a repeated position winning -- and the by-product
    dummy VCF byte for byte.  The thresholds handed to the kernels are strtof of exactly those cells (VC:887-890)."""
    from amplisolve_amd.hostio import ErrorTable

    table = _vc_table(name, tmp_path)
    t = ErrorTable(table, dummy_vcf=str(tmp_path / "dummy.vcf"))
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
    dig = _vc_digests()
    assert (hashlib.sha256(ours).hexdigest(), len(ours.splitlines())) == dig[f"{name}.maps"]
    if os.path.exists(f"{G}/vc_ref/{name}.maps.gz"):
        assert ours == gzip.open(f"{G}/vc_ref/{name}.maps.gz").read()
    vcf = (tmp_path / "dummy.vcf").read_bytes()
    assert (hashlib.sha256(vcf).hexdigest(), len(vcf.splitlines())) == dig[f"{name}.vcf"]
    # the numbers: strtof of the two halves of the reference's cell (sscanf "%[^_]_%[^_]", VC:886-890)
    step = max(1, t.P // 3000)
    for p in range(0, t.P, step):
        for nt in range(4):
            a, b = t.cell(p, 1 + nt).split("_")[:2]
            assert t.thr[0, nt, p].view(np.int32) == _strtof(a) and t.thr[1, nt, p].view(np.int32) == _strtof(b)


def test_error_table_reader_round_trip(tmp_path):
    """The same reader against the ORACLE's finalize (text round trip, 0.01 substitution): a self-consistency check beside
    the reference pin above."""
    d = f"{G}/toy_subset"
    co = HostCohort(f"{d}/panel.bed", f"{d}/NORMAL", refbases_file=f"{d}/refbases.txt")
    acc, fin, _ = oracle_table(co, 0.002, 100, tmp_path, "rt.txt")
    ref, thr = read_error_table(str(tmp_path / "rt.txt"))
    assert np.array_equal(ref, co.ref_code)
    for nt in range(4):
        m = co.ref_code != nt  # the ref cell is "-2_-2", never read by the caller
        assert np.array_equal(thr[:, nt][:, m].view(np.int32), fin["thr"][:, nt][:, m].view(np.int32))
        assert (thr[:, nt][:, ~m] == -2).all()


@pytest.mark.parametrize("name", list(VC_TABLES))
def test_sequence_context_against_the_reference_functions(tmp_path, name):
    """f3's context columns, reference-pinned: 10-mer down / up and the homopolymer flag of every table position x 4
    substituted bases == the reference's own find_kmer_down / find_kmer_up / homopolymerTest (VC:3307-3718) called the
    way callVariants calls them (VC:964-965, 1017)."""
    from amplisolve_amd.hostio import ErrorTable

    t = ErrorTable(_vc_table(name, tmp_path))
    want = gzip.open(f"{G}/vc_ref/{name}.context.gz").read().decode().splitlines()
    assert len(want) == t.P == _vc_digests()[f"{name}.context"][1]
    flagged = 0
    for p, w in enumerate(want):
        f = w.split(" ")
        c, x = t.key(p)
        assert (f[0], f[1], f[2]) == ("C", c, str(x))
        for i, sub in enumerate("ACGT"):
            down, up, flag = t.context(p, sub)
            assert (down, up, str(flag)) == (f[3], f[4], f[5 + i]), (w, sub)
            flagged += flag
    if name == "context_edge":
        assert flagged > 250  # the runs around the 18-of-21 decision are there
    if name == "toy_full_0.0020":
        assert flagged > 0


def test_tumour_visit_order_against_the_reference(tmp_path):
    """a1, VC twin (VC:387-394, 580-627): the order callVariants visits the tumour files in (VC:672) is the iteration order
    of the reference's own TumourFileList_Hash for the same directory literal."""
    want = [l.split("\t")[0] for l in open(f"{G}/vc_ref/toy_subset_TUMOUR.order").read().splitlines()]
    assert sample_order(f"{G}/toy_subset/TUMOUR") == want and sorted(want) == ["T1", "T2", "T3"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/Toy_data"), reason="full Toy_data only exists in the build container")
def test_full_toy_data_digest(tmp_path):
    """Config 1 of BASELINE.json: the whole Toy_data panel, digest of the reference's table."""
    T = "/root/reference/Toy_data"
    refb = tmp_path / "refbases.txt"
    refb.write_bytes(gzip.open(f"{G}/toy/refbases.txt.gz").read())
    co = HostCohort(f"{T}/AmpliSeq_30genes_Designed-1.bed", f"{T}/NORMAL_ASEQ_DIR", refbases_file=str(refb))
    assert co.names == open(f"{G}/toy/visit_order.txt").read().split()
    assert (co.P, co.walk_len, co.S) == (40814, 41486, 5)
    _, _, got = oracle_table(co, 0.002, 100, tmp_path)
    want = open(f"{G}/toy/digests.txt").read().split()[2]
    assert hashlib.sha256(got.encode()).hexdigest() == want == "ebb19204f4d1ca3020be3ee8b7457564c817f64c88166b11aa0f847ed826ba3a"
    assert got == gzip.open(f"{G}/toy/positionSpecificNoise_0.0020.txt.gz").read().decode()
    assert sample_order(f"{T}/TUMOUR_ASEQ_DIR") == ["T3", "T2", "T1"]


def test_scorer_against_reference_grid():
    z = np.load(f"{G}/vc_scorer_reference.npz")
    q, _ = orc.score_batch(z["k"], z["rd"], z["err"])
    assert np.array_equal(q.view(np.int64), z["q"].view(np.int64))  # bit for bit, incl. -888 / 100 / 0 specials
    L = orc.lib()
    gq = np.array([L.oracle_kf_gammaq(float(a), float(b)) for a, b in zip(z["s"], z["z"])])
    assert np.array_equal(gq.view(np.int64), z["gammaq"].view(np.int64))
    lg = np.array([L.oracle_kf_lgamma(float(a)) for a in z["s"]])
    assert np.array_equal(lg.view(np.int64), z["lgamma"].view(np.int64))


def test_scorer_known_answers_from_survey():
    """values the surveyor captured from the compiled reference."""
    L = orc.lib()
    f = float(np.float32(0.002))
    for s, zz, want in [(1, 1000 * f, 0.13533527038045126), (2, 1000 * f, 0.40600582399751517), (3, 1000 * f, 0.67667639047073891),
                        (5, 1000 * f, 0.94734697408551383), (8, 1000 * f, 0.99890328070563561), (40, 25000 * f, 0.064570328074039979)]:
        assert L.oracle_kf_gammaq(float(s), zz) == want
    for k, rd, e, want in [(0, 1000, 0.002, 0.0), (1, 1000, 0.002, 0.63152255889662207), (2, 1000, 0.002, 2.2621781317096674),
                           (3, 1000, 0.002, 4.9036258145739694), (5, 1000, 0.002, 12.785766653103605), (8, 1000, 0.002, 29.599045160064782),
                           (20, 1000, 0.002, 100.0), (250, 25000, 0.002, 100.0), (1, 100, 0.01, 1.9920009027716749),
                           (40, 25000, 0.002, 0.28988858343132076), (3, 1000, -1.0, -888.0), (3, 1000, 0.0, 10.944814144876279)]:
        assert float(L.oracle_score(k, rd, e)) == pytest.approx(want, rel=1e-15, abs=0)


@pytest.mark.skipif(not os.path.exists(orc.REF_VC_SCORER), reason="reference scorer build not present")
def test_scorer_bitwise_vs_reference_random():
    import ctypes as C

    R = C.CDLL(orc.REF_VC_SCORER)
    rng = np.random.default_rng(2)
    n = 500000
    k = rng.integers(0, 500, n).astype(np.int32)
    rd = rng.integers(0, 60000, n).astype(np.int32)
    err = rng.choice(np.array([0.002, 0.01, 0.0005, 0.05, 0.0, -1.0, 0.002189, 0.000123, 0.3], np.float32), n)
    want = np.empty(n)
    R.ref_score_batch(k.ctypes.data_as(C.c_void_p), rd.ctypes.data_as(C.c_void_p), err.ctypes.data_as(C.c_void_p), C.c_long(n),
                      want.ctypes.data_as(C.c_void_p))
    got, _ = orc.score_batch(k, rd, err)
    assert np.array_equal(got.view(np.int64), want.view(np.int64))


def test_irregular_lines_oracle_reproduces_the_reference_table():
    """Lines whose RD column is not A+C+G+T (a quarter of the lines of this fixture, in ways that move the AF <= 0.05 gate
    of Germ_Max both ways): the oracle, fed the RD column beside the records, writes the reference's table byte for byte
    (tests/golden/irregular, generated by make_golden_irregular.py from the compiled reference)."""
    d = os.path.join(G, "irregular")
    for tag, C_value, cov in (("0.0020_cov100", 0.002, 100), ("0.0100_cov1", 0.01, 1)):
        co = HostCohort(f"{d}/panel.bed", f"{d}/NORMAL", refbases_file=f"{d}/refbases.txt")
        assert co.names == open(f"{d}/expected_visit_order.txt").read().split()
        rd = co.rd_plane()
        assert rd is not None and (rd != np.iinfo(np.int32).min).sum() == co.stats()["irregular"] == 383
        acc = orc.error_reduce(co.recs, co.P, C_value, cov, E=co.E, dup_off=co.dup_off, rd=rd)
        fin = orc.error_finalize(acc)
        out = os.path.join(tempfile.mkdtemp(), "t.txt")
        co.write_error_table(fin["rate"], fin["code"], fin["germ_val"].astype(np.float32), fin["germ_present"], out)
        assert open(out).read() == open(f"{d}/expected_positionSpecificNoise_{tag}.txt").read()
        # and without the column the table is NOT the reference's: the fixture really exercises it
        acc0 = orc.error_reduce(co.recs, co.P, C_value, cov, E=co.E, dup_off=co.dup_off)
        assert not np.array_equal(acc0["gm_n"], acc["gm_n"])


@pytest.mark.parametrize("panel,arg,tag", [("mini_edge", "0.012", "0.0120"), ("mini_edge", "0.01", "0.0100"), ("mini_edge", "0", "0.0100"),
                                           ("mini_edge", "-3", "0.0100"), ("mini_edge", "junk", "0.0100"), ("mini_edge", "0.00049", "0.00049"),
                                           ("mini_edge", "0.123456", "0.123456"), ("mini_edge", "7", "7"), ("toy_subset", "0.012", "0.0120")])
def test_default_table_mode_is_byte_identical_to_the_reference(tmp_path, panel, arg, tag):
    """germline_dir=not_available (EE:472-506): the table generateFinalOutput_default (EE:2948-3043) itself wrote for the same
    panel (tests/golden/make_golden_default.py, the reference compiled where it lies) -- through the drop-in command line, which
    needs no GPU for this mode.  default_error <= 0 or unparsable becomes 0.01 in main() (EE:353-363: atof)."""
    import subprocess

    d = f"{G}/{panel}"
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "amplisolve_amd", "bin", "AmpliSolveErrorEstimation")
    out = tmp_path / "o"
    r = subprocess.run([exe, f"panel_design={d}/panel.bed", "reference_genome=unused.fa", "germline_dir=not_available", "C_value=0.002", "coverage_cutoff=100",
                        f"default_error={arg}", f"output_dir={out}"], capture_output=True, text=True,
                       env=dict(os.environ, AMPLISOLVE_STRICT_EXIT="1", AMPLISOLVE_REFBASES_FILE=f"{d}/refbases.txt"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert (out / "positionSpecificNoise_default.txt").read_bytes() == open(f"{d}/expected_positionSpecificNoise_default_{tag}.txt", "rb").read()
