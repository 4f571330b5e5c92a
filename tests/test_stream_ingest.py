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
