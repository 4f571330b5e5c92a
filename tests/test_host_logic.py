"""C++ host logic on the CPU: parsers, panel index, visit order, packer, table writer/reader edge cases
(the reference's behaviours cited per test).  No GPU, no oracle arithmetic."""
import os

import numpy as np
import pytest

from amplisolve_amd import AmpliError, host_lib
from amplisolve_amd.hostio import HostCohort, read_error_table, sample_order

ABSENT = np.iinfo(np.int32).min
HDR = "chr\tpos\tdbsnp\tMAF\tref\talt\tA\tC\tG\tT\tRD\tArs\tCrs\tGrs\tTrs\n"


def write_aseq(path, rows):
    with open(path, "w") as f:
        f.write(HDR)
        for r in rows:
            f.write("\t".join(str(x) for x in r) + "\n")


def test_bed_walk_crlf_duplicates_and_blank_lines(tmp_path):
    bed = tmp_path / "p.bed"
    bed.write_bytes(b"chr1\t10\t14\tA1\trs1\tG1\r\n\r\nchr1\t13\t16\tA2\trs2\tG2\r\nchrX\t5\t5\tA3\trs3\tG3\r\n")
    co = HostCohort(str(bed))
    # 1-based inclusive expansion (EE:633-649); chr1:13,14 listed twice -> duplicate YES (EE:657-664)
    assert co.P == 8 and co.walk_len == 10
    assert [co.position(p) for p in range(co.P)] == [("chr1", 10), ("chr1", 11), ("chr1", 12), ("chr1", 13), ("chr1", 14),
                                                      ("chr1", 15), ("chr1", 16), ("chrX", 5)]
    assert co.dup_flag.tolist() == [0, 0, 0, 1, 1, 0, 0, 0]


def test_reference_bases_from_fasta_with_and_without_fai(tmp_path):
    bed = tmp_path / "p.bed"
    bed.write_text("c1\t3\t6\tx\ty\tz\nc2\t1\t2\tx\ty\tz\nc3\t1\t1\tx\ty\tz\n")
    fa = tmp_path / "r.fa"
    fa.write_text(">c1 desc\nACG\nTac\nGT\n>c2\nNN\n")
    co = HostCohort(str(bed), fasta=str(fa))
    # case is preserved: lower-case (soft-masked) bases are not A/C/G/T for the reference (EE:2668-2670, VC:869)
    assert co.ref_code.tolist() == [2, 3, 255, 255, 255, 255, 255]  # G T a c | N N | contig c3 absent
    (tmp_path / "r.fa.fai").write_text("c1\t8\t9\t3\t4\nc2\t2\t24\t2\t3\n")
    co2 = HostCohort(str(bed), fasta=str(fa))
    assert co2.ref_code.tolist() == co.ref_code.tolist()


def test_aseq_packing_absent_offpanel_irregular_malformed_and_extras(tmp_path):
    bed = tmp_path / "p.bed"
    bed.write_text("chr1\t100\t103\ta\tb\tc\nchr1\t102\t104\ta\tb\tc\n")  # 102,103 listed twice
    d = tmp_path / "N"
    d.mkdir()
    write_aseq(d / "S1_x.PILEUP.ASEQ", [
        ("chr1", 100, ".", ".", ".", ".", 10, 0, 0, 90, 100, 4, 0, 0, 40),
        ("chr1", 102, ".", ".", ".", ".", 0, 50, 0, 0, 50, 0, 20, 0, 0),
        ("chr1", 102, ".", ".", ".", ".", 0, 51, 0, 0, 51, 0, 21, 0, 0),        # second listing -> extra record
        ("chr9", 999, ".", ".", ".", ".", 1, 1, 1, 1, 4, 0, 0, 0, 0),           # not on the panel: ignored
        ("chr1", 104, ".", ".", ".", ".", 1, 2, 3, 4, 11, 0, 1, 1, 2),           # RD != A+C+G+T: counted as irregular
        ("chr1", 101, ".", ".", ".", "."),                                      # malformed (too few fields)
    ])
    (d / "S2_y.PILEUP.ASEQ").write_text(HDR)  # header-only file (Toy_data's T3): a sample with no records
    co = HostCohort(str(bed), str(d), keep_line_no=True)
    assert co.S == 2 and co.P == 5 and co.E == 1 and sorted(co.names) == ["S1_x", "S2_y"]
    st = co.stats()
    assert st == dict(lines=6, offpanel=1, irregular=1, malformed=1)
    s1 = co.names.index("S1_x")
    assert co.recs[s1, 0].tolist() == [6, 0, 0, 50, 4, 0, 0, 40]           # fw = X - Xrs (EE:1155-1158)
    assert co.recs[s1, 1, 0] == ABSENT                                      # chr1:101 has no valid line
    assert co.recs[s1, 2].tolist() == [0, 30, 0, 0, 0, 20, 0, 0]
    assert co.dup_off.tolist() == [0, 0, 0, 1, 1, 1] and co.ext_pos.tolist() == [2]
    assert co.recs[s1, 5].tolist() == [0, 30, 0, 0, 0, 21, 0, 0]           # the extra occurrence of chr1:102
    assert co.line_no[s1].tolist() == [0, -1, 1, -1, 4, 2]
    assert (co.recs[1 - s1, :, 0] == ABSENT).all()


def test_sample_names_and_visit_order_alias(tmp_path):
    d = tmp_path / "dir"
    d.mkdir()
    for n in ("P1_T.PILEUP.ASEQ", "P2_T.PILEUP.ASEQ", "P3_T.PILEUP.ASEQ", "zz.PILEUP.ASEQ", "ignored.txt"):
        (d / n).write_text(HDR)
    names = sample_order(str(d))
    assert sorted(names) == ["P1_T", "P2_T", "P3_T", "zz"]           # name = file minus ".PILEUP.ASEQ" (EE:831)
    os.environ["AMPLISOLVE_LIST_DIR_AS"] = "/some/other/literal"
    try:
        alias = sample_order(str(d))
    finally:
        del os.environ["AMPLISOLVE_LIST_DIR_AS"]
    assert sorted(alias) == sorted(names)                                # same files, order replayed for another literal
    with pytest.raises(AmpliError):
        sample_order(str(tmp_path / "empty_or_missing"))


def test_error_table_writer_and_reader_cells(tmp_path):
    bed = tmp_path / "p.bed"
    bed.write_text("chr2\t7\t9\ta\tb\tc\nchr2\t9\t9\ta\tb\tc\n")
    refb = tmp_path / "ref.txt"
    refb.write_text("chr2\t7\tA\nchr2\t8\tg\nchr2\t9\tT\nchr2\t9\tT\n")
    co = HostCohort(str(bed), refbases_file=str(refb))
    P = co.P
    rate = np.zeros((2, 4, P), np.float32)
    rate[0, 1, 0], rate[1, 1, 0] = 0.002189, 0.0021670001
    code = np.zeros((4, P), np.uint8)
    code[2, 0] = 1   # below quorum -> 0.01_0.01 (EE:2680-2684)
    code[3, 1] = 2   # NaN -> 0.01_0.01
    gval = np.zeros((4, P), np.float32)
    gpres = np.zeros((4, P), np.uint8)
    gval[0, 1], gpres[0, 1] = -888, 1
    gval[1, 1], gpres[1, 1] = 0.00170648, 1
    out = tmp_path / "t.txt"
    co.write_error_table(rate, code, gval, gpres, str(out))
    rows = [l.split("\t") for l in out.read_text().splitlines()]
    assert rows[0][:5] == ["chrom", "position", "reference", "duplicate", "Thres_A"] and len(rows) == 1 + 4
    assert rows[1] == ["chr2", "7", "A", "NO", "-2_-2", "0.002189_0.002167", "0.01_0.01", "0.000000_0.000000", "-", "-", "-", "-"]
    assert rows[2][2:5] == ["g", "NO", "0.000000_0.000000"] and rows[2][7] == "0.01_0.01" and rows[2][8:10] == ["-888", "0.00170648"]
    assert rows[3][3] == "YES" and rows[4] == rows[3]                    # the duplicated position is written twice
    ref, thr = read_error_table(str(out))
    assert ref.tolist() == [0, 255, 3]                                   # lower-case g is not a callable reference
    assert thr[0, 1, 0] == np.float32(0.002189) and thr[1, 2, 0] == np.float32(0.01) and thr[0, 0, 0] == -2.0


def test_error_table_io_does_not_depend_on_the_thread_count(tmp_path, monkeypatch):
    """The table is formatted / tokenised in slices on up to 8 threads: 20 000 rows (duplicated positions included) written and read
    back with 1 and with 8 threads give the same bytes and the same thresholds."""
    rng = np.random.default_rng(8)
    bed = tmp_path / "p.bed"
    bed.write_text("".join(f"chr{1 + k % 5}\t{1000 * k + 1}\t{1000 * k + 40}\n" for k in range(490)) + "chr1\t20\t60\n")  # the last amplicon overlaps the first
    refb = tmp_path / "ref.txt"
    lines = []
    for k in range(490):
        lines += [f"chr{1 + k % 5}\t{p}\t{'ACGT'[p % 4]}\n" for p in range(1000 * k + 1, 1000 * k + 41)]
    lines += [f"chr1\t{p}\t{'ACGT'[p % 4]}\n" for p in range(20, 61)]
    refb.write_text("".join(lines))
    co = HostCohort(str(bed), refbases_file=str(refb))
    P = co.P
    rate = rng.random((2, 4, P)).astype(np.float32) * np.float32(0.01)
    code = (rng.random((4, P)) < 0.1).astype(np.uint8)
    gval = rng.random((4, P)).astype(np.float32)
    gpres = (rng.random((4, P)) < 0.5).astype(np.uint8)
    outs, thrs = [], []
    for n in ("1", "8"):
        monkeypatch.setenv("AMPLISOLVE_THREADS", n)
        out = tmp_path / f"t{n}.txt"
        co.write_error_table(rate, code, gval, gpres, str(out))
        outs.append(out.read_bytes())
        thrs.append(read_error_table(str(out))[1])
    assert outs[0] == outs[1] and outs[0].count(b"\n") == 1 + 490 * 40 + 41
    assert np.array_equal(thrs[0].view(np.int32), thrs[1].view(np.int32))


def test_error_table_writer_prints_like_printf(tmp_path):
    """The writer forms "%f" by exact integer arithmetic and "%g" through the C library (round 4, csrc/host/table.cpp put_f6):
    every cell must be what the reference's sprintf("%f_%f") (EE:1704) / `ostream << double` (EE:2807-2849) print -- checked
    against the C library's own formatting of the same floats: bit patterns drawn over the whole range (denormals, halves
    that round to even, negative values, values beyond 2^20 that take the fallback) and the error-rate range itself."""
    import ctypes as C

    libc = C.CDLL(None)
    libc.snprintf.restype = C.c_int
    bed = tmp_path / "p.bed"
    bed.write_text("chr1\t1\t30000\n")
    refb = tmp_path / "ref.txt"
    refb.write_text("".join(f"chr1\t{p}\tN\n" for p in range(1, 30001)))  # N: no cell is the "-2_-2" of a reference base
    co = HostCohort(str(bed), refbases_file=str(refb))
    P = co.P
    rng = np.random.default_rng(12)
    bits = rng.integers(0, 2**32, (2, 4, P), dtype=np.uint64).astype(np.uint32)
    rate = bits.view(np.float32).copy()
    rate[~np.isfinite(rate)] = np.float32(0.25)
    rate[:, :, : P // 3] = (rng.random((2, 4, P // 3)) * 0.06).astype(np.float32)              # the range rates live in
    k = rng.integers(0, 4000000, (2, 4, P // 3))
    rate[:, :, P // 3: 2 * (P // 3)] = ((2 * k + 1) / 2e6).astype(np.float32)                   # next to the ...5 ties of the 7th decimal
    rate[0, 0, :8] = np.array([0.0, -0.0, 1e-7, 5e-7, 1.5e-6, 2.5e-6, 1048575.9, 1048576.0], np.float32)
    gval = bits[0].view(np.float32).copy()
    gval[~np.isfinite(gval)] = np.float32(0.5)
    gval[:, : P // 2] = (rng.random((4, P // 2)) * 0.05).astype(np.float32)
    gval[0, :4] = np.array([0.0, -0.0, -888.0, 0.00170648], np.float32)
    out = tmp_path / "t.txt"
    co.write_error_table(rate, np.zeros((4, P), np.uint8), gval, np.ones((4, P), np.uint8), str(out))
    rows = out.read_text().splitlines()[1:]
    assert len(rows) == P
    buf = C.create_string_buffer(128)

    def fmt(f, v):
        libc.snprintf(buf, 128, f, C.c_double(float(v)))
        return buf.value.decode()

    for p in range(0, P, 1):
        cells = rows[p].split("\t")
        for nt in range(4):
            assert cells[4 + nt] == fmt(b"%f", rate[0, nt, p]) + "_" + fmt(b"%f", rate[1, nt, p]), (p, nt, rate[:, nt, p])
            assert cells[8 + nt] == fmt(b"%g", gval[nt, p]), (p, nt, gval[nt, p])


def test_fisher_matches_scipy():
    scipy = pytest.importorskip("scipy.stats")
    H = host_lib()
    rng = np.random.default_rng(5)
    for _ in range(300):
        a, b = int(rng.integers(100, 3000)), int(rng.integers(100, 3000))
        c, d = int(rng.integers(0, 60)), int(rng.integers(0, 60))
        # fisherTest(RD_fw, RD_bw, alt_fw, alt_bw): N=a+b+c+d, r=a+c, n=c+d (VC:3799-3803)
        got = H.ampli_host_fisher(a, b, c, d)
        want = scipy.fisher_exact([[a, c], [b, d]])[1]
        assert got == pytest.approx(want, rel=1e-6, abs=1e-12)
    assert H.ampli_host_fisher(461, 536, 196, 223) == pytest.approx(0.861148, abs=5e-7)  # SURVEY App. D, first Toy_data call


def test_fisher_recurrence_equals_the_term_by_term_sum():
    """ampli_host_fisher walks the pmf outwards from the mode by the ratio of neighbouring terms (round 4: the term-by-term
    log-gamma sum cost ~60-100 us per call at config-3 depths, the walk 3 us); it must print like the direct sum: equal to
    2e-9 relative -- the direct sum's own error at N ~ 2e5 is ~1e-10 per term (lgamma(N) ~ 1e6 carries an absolute error of
    ~2e-10), the walk's is smaller -- far inside the six digits of the Summary; small tables, deep ones, empty margins and
    the one-sided extremes."""
    H = host_lib()
    rng = np.random.default_rng(8)
    cases = [(0, 0, 0, 0), (5, 0, 0, 5), (0, 7, 3, 0), (1, 1, 1, 1), (3000, 2500, 2900, 10), (10, 12, 4000, 3900), (461, 536, 196, 223)]
    for _ in range(1500):
        hi = int(rng.choice([5, 60, 3000, 60000]))
        cases.append(tuple(int(x) for x in rng.integers(0, hi + 1, 4)))
    for _ in range(500):  # the shape the caller sees: depths and a smaller alt count per strand
        a, b = (int(x) for x in rng.integers(100, 30000, 2))
        cases.append((a, b, int(rng.integers(0, a + 1) * rng.random() ** 3), int(rng.integers(0, b + 1) * rng.random() ** 3)))
    for a, b, c, d in cases:
        got, want = H.ampli_host_fisher(a, b, c, d), H.ampli_host_fisher_direct(a, b, c, d)
        assert got == pytest.approx(want, rel=2e-9, abs=1e-300), (a, b, c, d, got, want)


def test_multi_front_end_token_parsing_and_usage(capsys):
    """amplisolve_amd/multi.py reads its tokens like the reference's mains (sscanf "key=%s", EE:300-326 / VC:242-260)."""
    from amplisolve_amd import multi

    assert multi._token("panel_design=/a/b.bed", "panel_design") == "/a/b.bed"
    assert multi._token("C_value=0.002 trailing", "C_value") == "0.002"
    assert multi._token("coverage=100", "coverage_cutoff") == "" and multi._token("p_value=", "p_value") == ""
    assert multi.main(["multi", "AmpliSolveVariantCalling", "errorFile=x"]) == 1  # wrong token count: usage, no GPU touched
    assert "usage" in capsys.readouterr().out


def test_cohort_shards_partition_the_visit_order():
    """ampli_host_cohort_load_shard: shard k of n holds exactly dist.shard_range(S, k, n) of the one-process cohort, record
    for record (what every rank of amplisolve_amd.multi parses)."""
    from amplisolve_amd.dist import shard_range

    d = "/root/repo/tests/golden/mini_edge"
    whole = HostCohort(f"{d}/panel.bed", f"{d}/NORMAL", refbases_file=f"{d}/refbases.txt")
    S = whole.S
    assert whole.first_sample == 0 and whole.total_samples == S and S >= 5
    for n in (2, 3, S + 2):
        seen = []
        for k in range(n):
            part = HostCohort(f"{d}/panel.bed", f"{d}/NORMAL", refbases_file=f"{d}/refbases.txt", shard=(k, n))
            lo, hi = shard_range(S, k, n)
            assert (part.first_sample, part.S, part.total_samples) == (lo, hi - lo, S)
            assert part.names == whole.names[lo:hi]
            if part.S:
                # the extras layout is per shard (it depends on the files seen): compare the primary records and every extra
                assert np.array_equal(part.recs[:, : whole.P], whole.recs[lo:hi, : whole.P])
                for s in range(part.S):
                    for p in range(whole.P):
                        a = part.recs[s, part.P + part.dup_off[p]: part.P + part.dup_off[p + 1]]
                        b = whole.recs[lo + s, whole.P + whole.dup_off[p]: whole.P + whole.dup_off[p + 1]]
                        na = [r.tolist() for r in a if r[0] != np.iinfo(np.int32).min]
                        nb = [r.tolist() for r in b if r[0] != np.iinfo(np.int32).min]
                        assert na == nb
            seen += part.names
        assert seen == whole.names
