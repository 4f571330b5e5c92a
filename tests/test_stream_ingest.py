"""The streaming ingest of the command lines (csrc/host/aseq.cpp: ChunkStream) on the CPU: chunks of consecutive samples
packed straight into the device record layout must hold exactly the records of the dense int32 array (the interchange
layout every oracle comparison is made on), whatever the chunk size; range validation; the int32 fall-back of a chunk."""
import os

import numpy as np
import pytest

from amplisolve_amd import AmpliError
from amplisolve_amd.hostio import HostCohort, unpack_records

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ABSENT = np.iinfo(np.int32).min
HEADER = "chr\tpos\tdbsnp\tMAF\tref\talt\tA\tC\tG\tT\tRD\tArs\tCrs\tGrs\tTrs\n"


def dense_from_chunks(chunks, P, S):
    """rebuild {sample: {(position, occurrence): record}} from the streamed chunks"""
    out = [dict() for _ in range(S)]
    lines = [dict() for _ in range(S)]
    for c in chunks:
        prim = unpack_records(c["prim"], c["layout"])
        ext = unpack_records(c["ext"], c["layout"]) if c["E"] else np.zeros((c["n"], 0, 8), np.int32)
        for i in range(c["n"]):
            s = c["first"] + i
            for p in np.nonzero(prim[i, :, 0] != ABSENT)[0]:
                out[s][(int(p), 0)] = prim[i, p]
                if c["line_prim"] is not None:
                    lines[s][(int(p), 0)] = int(c["line_prim"][i, p])
            for e in range(c["E"]):
                if ext[i, e, 0] != ABSENT:
                    p = int(c["ext_pos"][e])
                    k = e - int(c["dup_off"][p]) + 1
                    out[s][(p, k)] = ext[i, e]
                    if c["line_ext"] is not None:
                        lines[s][(p, k)] = int(c["line_ext"][i, e])
    return out, lines


def dense_from_cohort(co):
    out = [dict() for _ in range(co.S)]
    lines = [dict() for _ in range(co.S)]
    for s in range(co.S):
        for p in np.nonzero(co.recs[s, :co.P, 0] != ABSENT)[0]:
            out[s][(int(p), 0)] = co.recs[s, p]
            if co.line_no is not None:
                lines[s][(int(p), 0)] = int(co.line_no[s, p])
        for e in range(co.E):
            r = co.recs[s, co.P + e]
            if r[0] != ABSENT:
                p = int(co.ext_pos[e])
                k = e - int(co.dup_off[p]) + 1
                out[s][(p, k)] = r
                if co.line_no is not None:
                    lines[s][(p, k)] = int(co.line_no[s, co.P + e])
    return out, lines


@pytest.mark.parametrize("panel,aseq,chunk_bytes,threads", [("toy_subset", "NORMAL", 1 << 30, 0), ("toy_subset", "NORMAL", 1, 2), ("toy_subset", "TUMOUR", 100_000, 1),
                                                           ("mini_edge", "NORMAL", 1, 3), ("mini_edge", "NORMAL", 40_000, 0)])
def test_chunks_hold_the_records_of_the_dense_array(panel, aseq, chunk_bytes, threads):
    d = f"{G}/{panel}"
    co = HostCohort(f"{d}/panel.bed", f"{d}/{aseq}", refbases_file=f"{d}/refbases.txt", keep_line_no=True)
    chunks = co.stream_chunks(f"{d}/{aseq}", chunk_bytes=chunk_bytes, threads=threads, keep_line_no=True)
    assert sum(c["n"] for c in chunks) == co.S and [c["first"] for c in chunks] == list(np.cumsum([0] + [c["n"] for c in chunks[:-1]]))
    if chunk_bytes == 1 and threads:
        assert len(chunks) > 1  # really streamed in pieces
    assert all(c["layout"] == 1 for c in chunks)  # 16-byte records (every count <= 65534): what the command lines upload for such a cohort
    got, got_lines = dense_from_chunks(chunks, co.P, co.S)
    exp, exp_lines = dense_from_cohort(co)
    for s in range(co.S):
        assert got[s].keys() == exp[s].keys()
        for key in exp[s]:
            assert np.array_equal(got[s][key], exp[s][key]), (s, key)
        assert got_lines[s] == exp_lines[s]
    if panel == "mini_edge":
        assert co.E > 0 and any(c["E"] > 0 for c in chunks)


def _write_panel(tmp_path, lines_by_file):
    (tmp_path / "panel.bed").write_text("chr1\t100\t104\tA1\trs\tG\n")
    (tmp_path / "ref.txt").write_text("".join(f"chr1\t{p}\tA\n" for p in range(100, 105)))
    nd = tmp_path / "N"
    nd.mkdir()
    for name, lines in lines_by_file.items():
        (nd / f"{name}.PILEUP.ASEQ").write_text(HEADER + "".join(lines))
    return str(tmp_path / "panel.bed"), str(tmp_path / "ref.txt"), str(nd)


def _line(pos, A, C, G, T, rs, RD=None):
    RD = A + C + G + T if RD is None else RD
    return f"chr1\t{pos}\t.\t.\t.\t.\t{A}\t{C}\t{G}\t{T}\t{RD}\t{rs[0]}\t{rs[1]}\t{rs[2]}\t{rs[3]}\n"


def test_reverse_count_above_total_is_a_range_error(tmp_path):
    """X - Xrs < 0 is not a read count: the packer refuses it (AMPLI_E_RANGE) instead of handing a negative forward
    count to the kernels (include/amplisolve_hip.h)."""
    bed, ref, nd = _write_panel(tmp_path, {"S1": [_line(100, 500, 2, 1, 0, (250, 1, 0, 0)), _line(101, 10, 0, 0, 0, (11, 0, 0, 0))]})
    with pytest.raises(AmpliError) as e:
        HostCohort(bed, nd, refbases_file=ref)
    assert "negative" in str(e.value)
    co = HostCohort(bed, None, refbases_file=ref)
    with pytest.raises(AmpliError) as e:
        co.stream_chunks(nd)
    assert "(-6)" in str(e.value) and "negative" in str(e.value)  # AMPLI_E_RANGE


def test_a_count_beyond_the_layout_widens_its_chunk_and_at_most_the_next(tmp_path, monkeypatch):
    """The packer takes the narrowest layout a chunk's counts fit: uint16, 24 bits, int32.  A chunk is packed first in the
    layout the one before it needed, and again, wider, if a count does not fit; the chunk after an outlier is the only
    other one that may come out wider than it needs."""
    big = (1 << 24) + 5
    files = {f"S{i}": [_line(100 + j, 900 + i, 3, 2, 1, (450, 1, 1, 0)) for j in range(5)] for i in range(6)}
    files["S2"][3] = _line(103, big * 2, 3, 2, 1, (big, 1, 1, 0))       # beyond 24 bits
    files["S4"][1] = _line(101, 70_000 * 2, 3, 2, 1, (70_000, 1, 1, 0))  # beyond uint16, inside 24 bits
    bed, ref, nd = _write_panel(tmp_path, files)
    co = HostCohort(bed, nd, refbases_file=ref)
    chunks = co.stream_chunks(nd, chunk_bytes=1, threads=1)
    assert len(chunks) == 6
    order = [co.names[c["first"]] for c in chunks]
    layouts = [c["layout"] for c in chunks]
    need = {"S2": 0, "S4": 2}
    for i, (name, lay) in enumerate(zip(order, layouts)):
        want = need.get(name, 1)
        prev = need.get(order[i - 1], 1) if i else 1
        wide = {1: 0, 2: 1, 0: 2}
        assert lay == want or (wide[lay] > wide[want] and lay == prev), (order, layouts)
    assert layouts[order.index("S2")] == 0 and layouts[order.index("S4")] in (2, 0)
    monkeypatch.setenv("AMPLISOLVE_RECORDS", "u24")  # the narrowest the packer may choose
    assert all(c["layout"] in (2, 0) for c in co.stream_chunks(nd, chunk_bytes=1, threads=1))
    monkeypatch.delenv("AMPLISOLVE_RECORDS")
    got, _ = dense_from_chunks(chunks, co.P, co.S)
    exp, _ = dense_from_cohort(co)
    for s in range(co.S):
        for key in exp[s]:
            assert np.array_equal(got[s][key], exp[s][key])
    assert max(int(r.max()) for s in got for r in s.values()) == big


def test_irregular_lines_travel_as_a_side_list(tmp_path):
    """RD != A+C+G+T (EE:1178-1181): the record keeps its eight counts, the RD column travels beside it."""
    files = {"S1": [_line(100, 900, 3, 2, 1, (450, 1, 1, 0)), _line(101, 900, 3, 2, 1, (450, 1, 1, 0), RD=1000), _line(101, 800, 3, 2, 1, (400, 1, 1, 0), RD=7)],
             "S2": [_line(102, 700, 0, 0, 0, (300, 0, 0, 0))]}
    bed, ref, nd = _write_panel(tmp_path, files)
    co = HostCohort(bed, nd, refbases_file=ref)
    assert co.stats()["irregular"] == 2
    chunks = co.stream_chunks(nd, chunk_bytes=1 << 30)
    assert len(chunks) == 1 and chunks[0]["E"] == 1
    s1 = co.names.index("S1")
    irr = sorted(tuple(int(v) for v in r) for r in chunks[0]["irregular"])
    assert irr == [(s1, 1, 0, 1000), (s1, co.P + 0, 1, 7)]


def _chunks_signature(chunks):
    """everything a consumer of the stream sees, as bytes"""
    import hashlib

    h = hashlib.sha256()
    for c in chunks:
        h.update(repr((c["first"], c["n"], c["layout"], c["E"])).encode())
        for k in ("prim", "ext", "dup_off", "ext_pos", "irregular"):
            h.update(np.ascontiguousarray(c[k]).tobytes())
        for k in ("line_prim", "line_ext"):
            if c[k] is not None:
                h.update(np.ascontiguousarray(c[k]).tobytes())
    return h.hexdigest()


@pytest.mark.parametrize("panel", ["mini_edge", "irregular", "toy_subset"])
def test_binary_record_cache_gives_the_same_stream(tmp_path, monkeypatch, panel):
    """AMPLISOLVE_CACHE=1: the first pass over a directory leaves `<file>.aseqbin` beside every ASEQ file, the
    second reads those instead of the text -- and hands over the very same chunks (records in every layout, positions listed
    twice, lines with their own RD column, data-line indices, statistics).  A cache file is ignored when the text changed
    (size / mtime), when it was made against another panel, or when it is damaged; then the text is parsed and the cache rewritten."""
    import shutil

    src = f"{G}/{panel}"
    d = tmp_path / "N"
    shutil.copytree(f"{src}/NORMAL", d)
    bed, refb = f"{src}/panel.bed", f"{src}/refbases.txt"
    monkeypatch.delenv("AMPLISOLVE_CACHE", raising=False)
    co = HostCohort(bed, str(d), refbases_file=refb, keep_line_no=True)
    plain = {kl: _chunks_signature(co.stream_chunks(str(d), chunk_bytes=1, threads=2, keep_line_no=kl)) for kl in (True, False)}
    plain_stats = co.stats()
    assert not [f for f in os.listdir(d) if f.endswith(".aseqbin")]
    monkeypatch.setenv("AMPLISOLVE_CACHE", "1")
    first = _chunks_signature(co.stream_chunks(str(d), chunk_bytes=1, threads=2, keep_line_no=False))  # written by a pass that did not ask for line indices
    files = sorted(f for f in os.listdir(d) if f.endswith(".aseqbin"))
    assert len(files) == co.S and first == plain[False]
    stamp = {f: os.stat(d / f).st_mtime_ns for f in files}
    for kl in (True, False):
        for chunk_bytes, threads in ((1, 2), (1 << 30, 0)):
            got = co.stream_chunks(str(d), chunk_bytes=chunk_bytes, threads=threads, keep_line_no=kl)
            if chunk_bytes == 1:
                assert _chunks_signature(got) == plain[kl]
    assert {f: os.stat(d / f).st_mtime_ns for f in files} == stamp  # hits do not rewrite
    # the dense loader (interchange layout, what the oracle comparisons read) goes through the same cache
    co2 = HostCohort(bed, str(d), refbases_file=refb, keep_line_no=True)
    assert np.array_equal(co2.recs, co.recs) and np.array_equal(co2.line_no, co.line_no) and co2.stats() == plain_stats
    assert np.array_equal(co2.irregular, co.irregular)
    # wider layouts out of a uint16 cache: AMPLISOLVE_RECORDS sets the narrowest layout the packer may use
    for lay, code in (("u24", 2), ("i32", 0)):
        monkeypatch.setenv("AMPLISOLVE_RECORDS", lay)
        monkeypatch.setenv("AMPLISOLVE_CACHE", "0")
        want = co.stream_chunks(str(d), chunk_bytes=1 << 30, keep_line_no=True)
        monkeypatch.setenv("AMPLISOLVE_CACHE", "1")
        got = co.stream_chunks(str(d), chunk_bytes=1 << 30, keep_line_no=True)
        assert all(c["layout"] == code for c in got) and _chunks_signature(got) == _chunks_signature(want)
    monkeypatch.delenv("AMPLISOLVE_RECORDS")
    # invalidation: a changed text file, a damaged cache file, another panel
    victim = sorted(f for f in os.listdir(d) if f.endswith(".ASEQ"))[0]
    text = (d / victim).read_text().splitlines(keepends=True)
    (d / victim).write_text("".join(text[:-1]))  # one data line less
    co3 = HostCohort(bed, str(d), refbases_file=refb, keep_line_no=True)
    assert co3.stats()["lines"] == plain_stats["lines"] - 1
    assert os.stat(d / (victim + ".aseqbin")).st_mtime_ns != stamp[victim + ".aseqbin"]  # rewritten against the new text
    other = sorted(files)[-1]
    blob = (d / other).read_bytes()
    (d / other).write_bytes(blob[: len(blob) // 2])
    co4 = HostCohort(bed, str(d), refbases_file=refb, keep_line_no=True)
    assert np.array_equal(co4.recs, co3.recs) and len((d / other).read_bytes()) == len(blob)
    bed2 = tmp_path / "shifted.bed"
    bed2.write_text("".join(l for i, l in enumerate(open(bed).read().splitlines(keepends=True)) if i != 1))  # one amplicon less: other positions
    monkeypatch.setenv("AMPLISOLVE_CACHE", "0")
    want = HostCohort(str(bed2), str(d), refbases_file=refb, keep_line_no=True)
    monkeypatch.setenv("AMPLISOLVE_CACHE", "1")
    got = HostCohort(str(bed2), str(d), refbases_file=refb, keep_line_no=True)
    assert got.P == want.P != co.P and np.array_equal(got.recs, want.recs) and np.array_equal(got.line_no, want.line_no)


def test_binary_record_cache_when_a_file_needs_wider_records(tmp_path, monkeypatch):
    """A file whose counts need 24-bit fields, cached, then asked for in a chunk that is being packed as uint16: the cache answers
    like the text parser (the chunk is packed again, wider), and the streams with and without the cache are the same bytes."""
    files = {f"S{i}": [_line(p, 1000 + i, 2, 1, 3, (500 + i, 1, 0, 1)) for p in range(100, 105)] for i in range(5)}
    files["S3"][2] = _line(102, 900_000, 3, 1, 2, (400_000, 1, 0, 1))  # beyond uint16
    bed, refb, d = _write_panel(tmp_path, files)
    monkeypatch.delenv("AMPLISOLVE_CACHE", raising=False)
    co = HostCohort(bed, d, refbases_file=refb, keep_line_no=True)
    want = co.stream_chunks(d, chunk_bytes=1, threads=1, keep_line_no=True)
    monkeypatch.setenv("AMPLISOLVE_CACHE", "1")
    made = co.stream_chunks(d, chunk_bytes=1, threads=1, keep_line_no=True)
    again = co.stream_chunks(d, chunk_bytes=1, threads=1, keep_line_no=True)
    assert _chunks_signature(made) == _chunks_signature(want) == _chunks_signature(again)
    assert sorted({c["layout"] for c in again}) == [1, 2]  # uint16 chunks and the 24-bit one
    one = co.stream_chunks(d, chunk_bytes=1 << 30, keep_line_no=True)
    assert len(one) == 1 and one[0]["layout"] == 2


def _hostile_files(rng, positions, n_files):
    """ASEQ files whose lines stray from `tab, unsigned integer, tab` in every way the plain tokeniser has an opinion on"""
    def clean(chrom, pos):
        A, C, G, T = (int(x) for x in rng.integers(0, 3000, 4))
        rs = [int(rng.integers(0, v + 1)) for v in (A, C, G, T)]
        RD = A + C + G + T + (int(rng.integers(1, 9)) if rng.random() < 0.05 else 0)
        return [chrom, str(pos), "rs1", "0.1", "A", "G", str(A), str(C), str(G), str(T), str(RD)] + [str(x) for x in rs]

    files = {}
    for f in range(n_files):
        out = []
        for chrom, pos in positions:
            r = rng.random()
            if r < 0.1:
                continue  # position absent from this file
            for _ in range(1 + (rng.random() < 0.1)):  # some positions listed twice
                tok = clean(chrom, pos)
                sep, eol = "\t", "\n"
                m = rng.random()
                if m < 0.45:
                    pass
                elif m < 0.50:
                    sep = " "
                elif m < 0.54:
                    sep = "\t\t"
                elif m < 0.58:
                    eol = "\r\n"
                elif m < 0.61:
                    tok[0] = " " + tok[0]  # leading blank
                elif m < 0.64:
                    i = int(rng.integers(6, 15))
                    tok[i] = "+" + tok[i]
                elif m < 0.67:
                    tok[int(rng.integers(6, 15))] += ".5"  # a float: the digits, then ".5" starts the next token
                elif m < 0.70:
                    tok += ["extra", "columns 1 2"]
                elif m < 0.73:
                    tok[-1] += "   "
                elif m < 0.76:
                    tok = tok[: int(rng.integers(1, 15))]  # short line
                elif m < 0.79:
                    tok[int(rng.integers(6, 15))] = "1" * 18  # beyond the tokeniser's 17 digits
                elif m < 0.82:
                    tok[10] = "0" * 12 + "7"  # RD with leading zeros, 13 digits: a line with its own RD column
                elif m < 0.85:
                    tok[0] = "chrUn"  # a chromosome the panel does not have
                elif m < 0.88:
                    tok[1] = str(pos + 100000)  # a coordinate the panel does not have
                elif m < 0.90:
                    tok[int(rng.integers(6, 15))] = "x7"
                elif m < 0.92:
                    tok[1] = tok[1] + "abc"
                elif m < 0.94:
                    tok[int(rng.integers(2, 6))] = ""  # an empty column: two tabs in a row
                elif m < 0.96:
                    tok[-1] += "\t"
                elif m < 0.98:
                    tok[-1] += "\r"  # a carriage return that is not the line's end
                    eol = " \n"
                else:
                    tok[-1] = "000" + tok[-1] + "abc"
                out.append(sep.join(tok) + eol)
                b = rng.random()
                if b < 0.03:
                    out.append("\n")
                elif b < 0.05:
                    out.append(" \t \r\n")
                elif b < 0.06:
                    out.append("\r\n")
        text = HEADER + "".join(out)
        if f % 3 == 1:
            text = text.rstrip("\n")  # the last line has no newline
        files[f"H{f}"] = text
    files["Hempty"] = ""
    files["Hheader"] = HEADER
    files["Hnonl"] = HEADER.rstrip("\n")
    files["Hblank"] = HEADER + "\n\n \n"
    return files


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fast_tokeniser_reads_hostile_files_exactly_as_the_plain_one(tmp_path, monkeypatch, seed):
    """The one-pass tokeniser of csrc/host/aseq.cpp only accepts lines it provably reads as the plain one (memchr + whitespace runs)
    does and hands every other line to it: same records, extras, RD side list, data-line indices and line statistics on files full
    of blanks, double tabs, carriage returns, signs, floats, short lines, over-long integers, extra columns and missing final newlines."""
    rng = np.random.default_rng(seed)
    positions = [("chr1", 1000 + i) for i in range(120)] + [("chr2", 5 + 3 * i) for i in range(60)]
    (tmp_path / "panel.bed").write_text("chr1\t1000\t1119\tA1\trs\tG\nchr2\t5\t182\tA2\trs\tG\n")
    (tmp_path / "ref.txt").write_text("".join(f"chr1\t{p}\tA\n" for p in range(1000, 1120)) + "".join(f"chr2\t{p}\tC\n" for p in range(5, 183)))
    nd = tmp_path / "N"
    nd.mkdir()
    for name, text in _hostile_files(rng, positions, 7).items():
        (nd / f"{name}.PILEUP.ASEQ").write_bytes(text.encode())
    bed, ref = str(tmp_path / "panel.bed"), str(tmp_path / "ref.txt")
    seen = {}
    for parser in ("plain", "fast"):
        if parser == "plain":
            monkeypatch.setenv("AMPLISOLVE_PARSER", "plain")
        else:
            monkeypatch.delenv("AMPLISOLVE_PARSER")
        co = HostCohort(bed, str(nd), refbases_file=ref, keep_line_no=True)
        sig = [_chunks_signature(co.stream_chunks(str(nd), chunk_bytes=cb, threads=t, keep_line_no=True)) for cb, t in ((1, 1), (1 << 30, 3))]
        seen[parser] = (co.stats(), co.recs.tobytes(), co.line_no.tobytes(), co.E, sig)
        co.close()
    assert seen["plain"][0] == seen["fast"][0], (seen["plain"][0], seen["fast"][0])
    assert seen["plain"][0]["malformed"] > 20 and seen["plain"][0]["lines"] > 900  # the hostile lines are there
    assert seen["plain"] == seen["fast"]


def test_both_tokenisers_refuse_the_same_line_with_the_same_words(tmp_path, monkeypatch):
    files = {"S1": [_line(100, 500, 2, 1, 0, (250, 1, 0, 0)), "\n", _line(101, 10, 0, 0, 0, (5, 0, 0, 0)).replace("\t", " ", 1), _line(102, 10, 0, 0, 0, (11, 0, 0, 0)),
                    _line(103, 10, 0, 0, 0, (1, 0, 0, 0))]}
    bed, ref, nd = _write_panel(tmp_path, files)
    said = []
    for parser in ("plain", None):
        monkeypatch.setenv("AMPLISOLVE_PARSER", parser) if parser else monkeypatch.delenv("AMPLISOLVE_PARSER")
        with pytest.raises(AmpliError) as e:
            HostCohort(bed, nd, refbases_file=ref)
        said.append(str(e.value))
    assert said[0] == said[1] and "data line 3:" in said[0] and "negative" in said[0]
