"""computeCounts on the GPU (BAM -> .PILEUP.ASEQ; device-side record decoding + counting, ampli_pileup_count) against the
Python restatement of the pileup step on the same synthetic BAM files, byte for byte."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import pileup_oracle as po
from tests import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "amplisolve_amd", "bin")
REFS = [("chr1", 100000), ("chr8", 50000), ("chrX", 30000), ("chrUn", 5000)]
AMPS = [(0, 1000, 1120), (0, 1100, 1230), (1, 5000, 5100), (2, 200, 330), (2, 29950, 29999)]


def _positions(rng):
    """the VCF-like list: every position of the amplicons in order (overlaps -> positions listed twice), plus a chromosome the
    BAM does not know and an uncovered stretch"""
    lines = []
    names = [n for n, _ in REFS]
    for ref_id, s, e in AMPS:
        lines += [(names[ref_id], p, ".", ".", ".") for p in range(s, e + 1)]
    lines += [("chr22", p, ".", ".", ".") for p in range(10, 15)] + [("chr8", p, "rs1", "A", "G") for p in range(40000, 40005)]
    return lines


def _write_vcf(path, lines):
    with open(path, "w") as f:
        for c, p, i, r, a in lines:
            f.write(f"{c}\t{p}\t{i}\t{r}\t{a}\t.\t.\t.\n")


@pytest.mark.parametrize("seed,n_reads,mbq,mrq,mdc,batch", [(1, 4000, 20, 20, 20, None), (2, 1500, 0, 0, 0, "70000"), (3, 6000, 21, 40, 1, "70000"),
                                                           (4, 300, 30, 5, 50, None)])
def test_compute_counts_equals_the_restatement(tmp_path, seed, n_reads, mbq, mrq, mdc, batch):
    rng = np.random.default_rng(seed)
    reads = helpers.random_amplicon_reads(rng, REFS, AMPS, n_reads)
    helpers.write_bam(tmp_path / "S1.bam", REFS, reads, rng=rng, max_block=20000)
    lines = _positions(rng)
    _write_vcf(tmp_path / "v.txt", lines)
    out = tmp_path / "out"
    out.mkdir()
    env = dict(os.environ)
    if batch:
        env["AMPLISOLVE_BAM_BATCH_BYTES"] = batch  # many small batches: records and even the header span batch boundaries
    r = subprocess.run([f"{BIN}/computeCounts", f"vcf={tmp_path}/v.txt", f"bam={tmp_path}/S1.bam", "threads=3", f"mbq={mbq}", f"mrq={mrq}", f"mdc={mdc}",
                        f"out={out}"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    refs, recs = po.read_bam(tmp_path / "S1.bam")
    want = po.aseq_text(lines, po.pileup(refs, recs, [(c, p) for c, p, *_ in lines], mbq, mrq), mdc)
    got = (out / "S1.PILEUP.ASEQ").read_text()
    assert got == want
    if mdc <= 1:
        assert len(got.splitlines()) > 300


def test_pileup_counts_feed_the_error_estimation(tmp_path):
    """The step's output is the next step's input: three BAM files -> three ASEQ files -> AmpliSolveErrorEstimation runs on them."""
    rng = np.random.default_rng(9)
    lines = [(n, p, ".", ".", ".") for n, s, e in (("chr1", 1000, 1120), ("chr8", 5000, 5100)) for p in range(s, e + 1)]
    _write_vcf(tmp_path / "v.txt", lines)
    (tmp_path / "N").mkdir()
    for k in range(3):
        reads = [dict(ref_id=rid, pos=s - 1, mapq=60, flag=0x10 * int(rng.integers(2)), cigar=[("M", e - s + 1)],
                      seq="".join(rng.choice(list("ACGT"), p=[0.94, 0.02, 0.02, 0.02], size=e - s + 1)), qual=[35] * (e - s + 1))
                 for rid, s, e in ((0, 1000, 1120), (1, 5000, 5100)) for _ in range(400)]
        helpers.write_bam(tmp_path / f"N{k}.bam", REFS, reads, rng=rng)
        r = subprocess.run([f"{BIN}/computeCounts", f"vcf={tmp_path}/v.txt", f"bam={tmp_path}/N{k}.bam", "mbq=20", "mrq=20", "mdc=20", f"out={tmp_path}/N"],
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
    (tmp_path / "p.bed").write_text("chr1\t1000\t1120\nchr8\t5000\t5100\n")
    (tmp_path / "ref.txt").write_text("".join(f"{c} {p} A\n" for c, p, *_ in lines))
    r = subprocess.run([f"{BIN}/AmpliSolveErrorEstimation", f"panel_design={tmp_path}/p.bed", "reference_genome=unused.fa", f"germline_dir={tmp_path}/N",
                        "C_value=0.002", "coverage_cutoff=100", "default_error=0.01", f"output_dir={tmp_path}/ee"], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, AMPLISOLVE_STRICT_EXIT="1", AMPLISOLVE_REFBASES_FILE=f"{tmp_path}/ref.txt"))
    assert r.returncode == 0, r.stdout + r.stderr
    table = (tmp_path / "ee" / "positionSpecificNoise_0.0020.txt").read_text().splitlines()
    assert len(table) == 1 + len(lines)


def test_pileup_count_entry_point_checks_its_arguments(ctx):
    assert ctx.lib.ampli_pileup_count(ctx.h, None, None, 1, None, 1, 20, 20, None, None) != 0
    assert b"bad argument" in ctx.lib.ampli_last_error(ctx.h)


def test_unsorted_reads_long_reads_and_an_empty_file(tmp_path):
    """The LDS window of the counting kernel assumes nothing: reads in random order (every update outside the window of its
    workgroup goes to the global counters), a read whose match run is longer than the window, and a BAM with a header and no
    alignment at all."""
    rng = np.random.default_rng(12)
    reads = helpers.random_amplicon_reads(rng, REFS, AMPS, 3000)
    reads.append(dict(ref_id=0, pos=900, mapq=60, flag=0, cigar=[("M", 1500)], seq="".join(rng.choice(list("ACGT"), size=1500)), qual=[40] * 1500))
    order = rng.permutation(len(reads))
    reads = [reads[i] for i in order]
    helpers.write_bam(tmp_path / "U.bam", REFS, reads, rng=rng, max_block=30000)
    lines = _positions(rng) + [("chr1", p, ".", ".", ".") for p in range(1231, 2400)]
    _write_vcf(tmp_path / "v.txt", lines)
    r = subprocess.run([f"{BIN}/computeCounts", f"vcf={tmp_path}/v.txt", f"bam={tmp_path}/U.bam", "mbq=20", "mrq=20", "mdc=1", f"out={tmp_path}"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    refs, recs = po.read_bam(tmp_path / "U.bam")
    want = po.aseq_text(lines, po.pileup(refs, recs, [(c, p) for c, p, *_ in lines], 20, 20), 1)
    assert (tmp_path / "U.PILEUP.ASEQ").read_text() == want
    helpers.write_bam(tmp_path / "E.bam", REFS, [])
    r = subprocess.run([f"{BIN}/computeCounts", f"vcf={tmp_path}/v.txt", f"bam={tmp_path}/E.bam", "mdc=0", f"out={tmp_path}"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = (tmp_path / "E.PILEUP.ASEQ").read_text().splitlines()
    assert len(got) == 1 + len(lines) and all(l.split("\t")[6:] == ["0"] * 9 for l in got[1:])  # mdc = 0: every listed position, all zeros
    r = subprocess.run([f"{BIN}/computeCounts", f"vcf={tmp_path}/v.txt", f"bam={tmp_path}/missing.bam", f"out={tmp_path}"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot open" in r.stdout


def test_long_reads_take_the_unstaged_walk(tmp_path):
    """256 consecutive records of 300-500 bp reads do not fit the kernel's 56 KB LDS stage: those groups are walked one wave per
    read from global memory (the fallback of pileup_count_staged_kernel); same counts."""
    rng = np.random.default_rng(21)
    amps = [(0, 1000, 1600), (1, 5000, 5700)]
    reads = helpers.random_amplicon_reads(rng, REFS, amps, 1200, read_len=(300, 500))
    helpers.write_bam(tmp_path / "L.bam", REFS, reads, rng=rng)
    lines = [(REFS[r][0], p, ".", ".", ".") for r, s, e in amps for p in range(s, e + 1)]
    _write_vcf(tmp_path / "v.txt", lines)
    r = subprocess.run([f"{BIN}/computeCounts", f"vcf={tmp_path}/v.txt", f"bam={tmp_path}/L.bam", "mbq=20", "mrq=20", "mdc=1", f"out={tmp_path}"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    refs, recs = po.read_bam(tmp_path / "L.bam")
    assert sum(len(x["seq"]) for x in recs[:256]) > 56 * 1024
    want = po.aseq_text(lines, po.pileup(refs, recs, [(c, p) for c, p, *_ in lines], 20, 20), 1)
    assert (tmp_path / "L.PILEUP.ASEQ").read_text() == want


def test_corrupt_bam_ends_in_an_error_and_no_counts_file(tmp_path):
    """A block_size gone wrong in the middle of the file: computeCounts names the place, exits non-zero under
    AMPLISOLVE_STRICT_EXIT and writes no .PILEUP.ASEQ (round 2 wrote one from the reads before the damage and exited 0)."""
    rng = np.random.default_rng(21)
    reads = helpers.random_amplicon_reads(rng, REFS, AMPS, 800)
    helpers.write_bam(tmp_path / "C.bam", REFS, reads, rng=rng, max_block=5000, corrupt_block_size=(400, 0x7FFFFFFF))
    _write_vcf(tmp_path / "v.txt", _positions(rng))
    r = subprocess.run([f"{BIN}/computeCounts", f"vcf={tmp_path}/v.txt", f"bam={tmp_path}/C.bam", "mbq=20", "mrq=20", "mdc=0", f"out={tmp_path}"],
                       capture_output=True, text=True, timeout=300, env=dict(os.environ, AMPLISOLVE_STRICT_EXIT="1"))
    assert r.returncode != 0 and "corrupt BAM record at uncompressed byte" in r.stdout
    assert not os.path.exists(tmp_path / "C.PILEUP.ASEQ")
