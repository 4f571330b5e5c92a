"""The pileup step upstream of the path (computeCounts: BAM -> .PILEUP.ASEQ), host side and oracle: no GPU needed.
The oracle (oracle/pileup_oracle.py) is pinned by hand-computed columns and by the observable properties of the reference's
own outputs of this step (Toy_data/*.PILEUP.ASEQ); the BGZF / BAM reader of the host library by a writer
that shares no code with it."""
import ctypes as C
import os

import numpy as np
import pytest

from amplisolve_amd import host_lib
from oracle import pileup_oracle as po
from tests import helpers

REFS = [("chr1", 100000), ("chr8", 50000), ("chrX", 30000)]


def _read(ref_id, pos, cigar, seq, qual=None, flag=0, mapq=60):
    return dict(ref_id=ref_id, pos=pos, mapq=mapq, flag=flag, cigar=cigar, seq=seq, qual=qual or [30] * len(seq))


def test_oracle_known_answer_columns(tmp_path):
    """Three reads over chr1:101-110, every rule of the restatement once: soft clip, insertion, deletion, reverse strand, low base
    quality, N, a filtered duplicate and a low-MAPQ read."""
    reads = [
        _read(0, 100, [("S", 2), ("M", 4), ("I", 1), ("M", 3)], "TTACGTAGGA"),           # 101 A 102 C 103 G 104 T | ins A | 105 G 106 G 107 A
        _read(0, 102, [("M", 2), ("D", 3), ("M", 3)], "GNCAT", qual=[30, 30, 5, 30, 30], flag=0x10),  # 103 G 104 N | del 105-107 | 108 C(q5) 109 A 110 T
        _read(0, 100, [("M", 5)], "AAAAA", flag=0x400),                                    # duplicate: ignored
        _read(0, 100, [("M", 5)], "CCCCC", mapq=3),                                        # MAPQ < 20: ignored
    ]
    helpers.write_bam(tmp_path / "k.bam", REFS, reads)
    refs, recs = po.read_bam(tmp_path / "k.bam")
    got = po.pileup(refs, recs, [("chr1", p) for p in range(100, 112)], mbq=20, mrq=20)
    want = {101: [1, 0, 0, 0, 0, 0, 0, 0], 102: [0, 1, 0, 0, 0, 0, 0, 0], 103: [0, 0, 2, 0, 0, 0, 1, 0], 104: [0, 0, 0, 1, 0, 0, 0, 0],
            105: [0, 0, 1, 0, 0, 0, 0, 0], 106: [0, 0, 1, 0, 0, 0, 0, 0], 107: [1, 0, 0, 0, 0, 0, 0, 0], 108: [0] * 8,
            109: [1, 0, 0, 0, 1, 0, 0, 0], 110: [0, 0, 0, 1, 0, 0, 0, 1], 100: [0] * 8, 111: [0] * 8}
    assert {p: got[("chr1", p)] for p in want} == want
    text = po.aseq_text([("chr1", 103, ".", ".", "."), ("chr1", 108, ".", ".", "."), ("chr1", 103, ".", ".", ".")], got, mdc=1)
    assert text.splitlines() == ["chr\tpos\tdbsnp\tMAF\tref\talt\tA\tC\tG\tT\tRD\tArs\tCrs\tGrs\tTrs",
                                 "chr1\t103\t.\t.\t.\t.\t0\t0\t2\t0\t2\t0\t0\t1\t0", "chr1\t103\t.\t.\t.\t.\t0\t0\t2\t0\t2\t0\t0\t1\t0"]


@pytest.mark.skipif(not os.path.isdir("/root/reference/Toy_data"), reason="reference fixtures are not on this machine")
def test_oracle_format_has_the_properties_of_the_reference_outputs():
    """What the reference's own outputs of this step show (there is no input BAM to go further): header, 15 tab-separated columns,
    RD = A+C+G+T, reverse counts within the totals, positions listed twice written twice with identical counts."""
    path = "/root/reference/Toy_data/NORMAL_ASEQ_DIR/N1.PILEUP.ASEQ"
    lines = open(path).read().splitlines()
    assert lines[0] == po.aseq_text([], {}, 0).strip()
    seen = {}
    twice = 0
    for l in lines[1:]:
        f = l.split("\t")
        assert len(f) == 15
        a = [int(x) for x in f[6:]]
        assert a[4] == sum(a[:4]) and all(r <= t for r, t in zip(a[5:], a[:4]))
        if (f[0], f[1]) in seen:
            assert seen[(f[0], f[1])] == a
            twice += 1
        seen[(f[0], f[1])] = a
    assert twice > 600


def test_host_reads_what_an_independent_writer_wrote(tmp_path):
    rng = np.random.default_rng(5)
    reads = helpers.random_amplicon_reads(rng, REFS, [(0, 1000, 1120), (1, 5000, 5100), (2, 200, 330)], 2500)
    n_bytes = helpers.write_bam(tmp_path / "a.bam", REFS, reads, rng=rng, max_block=3000)  # ~250 small blocks, records span them
    st = (C.c_int64 * 4)()
    assert host_lib().ampli_host_bam_scan(str(tmp_path / "a.bam").encode(), 3, st) == 0
    assert list(st) == [len(reads), n_bytes, len(REFS), 0]
    refs, recs = po.read_bam(tmp_path / "a.bam")
    assert refs == [n for n, _ in REFS] and len(recs) == len(reads)


def test_host_drops_malformed_records_and_rejects_broken_files(tmp_path):
    good = _read(0, 10, [("M", 5)], "ACGTA")
    bad = _read(0, 10, [("M", 7)], "ACGTA")  # the CIGAR consumes 7 bases, the read has 5: the kernel must never see it
    helpers.write_bam(tmp_path / "m.bam", REFS, [good, bad, good])
    st = (C.c_int64 * 4)()
    assert host_lib().ampli_host_bam_scan(str(tmp_path / "m.bam").encode(), 1, st) == 0
    assert st[0] == 2 and st[3] == 1
    raw = open(tmp_path / "m.bam", "rb").read()
    (tmp_path / "t.bam").write_bytes(raw[: len(raw) // 2])
    assert host_lib().ampli_host_bam_scan(str(tmp_path / "t.bam").encode(), 1, st) != 0
    # a block_size gone wrong in the middle of the file is a corrupt stream, not "a record that continues in the next batch"
    # (that reading carried the rest of the file along and ended in a silently truncated pileup): the reader says where and fails
    rng = np.random.default_rng(3)
    reads = helpers.random_amplicon_reads(rng, REFS, [(0, 1000, 1120)], 400)
    helpers.write_bam(tmp_path / "c.bam", REFS, reads, rng=rng, max_block=3000, corrupt_block_size=(200, 0x7FFFFFFF))
    assert host_lib().ampli_host_bam_scan(str(tmp_path / "c.bam").encode(), 2, st) != 0
    assert b"corrupt BAM record at uncompressed byte" in host_lib().ampli_host_last_error()
    # ... and one that is merely too short leaves the stream out of step: whatever follows, the file does not end on a record boundary
    helpers.write_bam(tmp_path / "d.bam", REFS, reads[:50], rng=rng, max_block=3000, corrupt_block_size=(49, 31))
    rc = host_lib().ampli_host_bam_scan(str(tmp_path / "d.bam").encode(), 2, st)
    assert rc != 0 or st[3] >= 1
    (tmp_path / "n.bam").write_bytes(b"this is not a BAM file, not even gzip" * 3)
    assert host_lib().ampli_host_bam_scan(str(tmp_path / "n.bam").encode(), 1, st) != 0
    assert b"BGZF" in host_lib().ampli_host_last_error()
    assert host_lib().ampli_host_bam_scan(str(tmp_path / "missing.bam").encode(), 1, st) != 0


def test_compute_counts_needs_the_gpu(tmp_path):
    """The product path has no CPU fallback: without a device computeCounts fails loudly."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    helpers.write_bam(tmp_path / "g.bam", REFS, [_read(0, 10, [("M", 5)], "ACGTA")])
    (tmp_path / "v.txt").write_text("chr1\t12\t.\t.\t.\t.\t.\t.\n")
    rc = host_lib().ampli_host_compute_counts(str(tmp_path / "v.txt").encode(), str(tmp_path / "g.bam").encode(), str(tmp_path).encode(), 2, 20, 20, 1, None)
    assert rc != 0 and not os.path.exists(tmp_path / "g.PILEUP.ASEQ")
