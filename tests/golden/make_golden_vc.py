#!/usr/bin/env python3
"""Golden fixtures for the table reader and the sequence-context columns from the REFERENCE ITSELF (round 4).

Run in the build container only (needs /root/reference and oracle/_ref, i.e. `make -C oracle`), from /root/repo:
    python tests/golden/make_golden_vc.py

oracle/_ref/vc_ref_driver = the Boost-free lines of AmpliSolveVariantCalling.cpp compiled where they lie (oracle/Makefile,
VC_HOST_SLICE; no stand-in for Boost, callVariants / fisherTest are not part of it) + oracle/ref_vc_driver.cpp.  For every
table listed below it runs the reference's own
  * storeInputFile (VC:430-576)                     -> <name>.maps    the four maps it fills, sorted by key
                                                    -> <name>.vcf     the by-product dummy VCF it writes
  * find_kmer_down / find_kmer_up / homopolymerTest (VC:3307-3718), called as callVariants calls them (VC:964-965, 1017)
                                                    -> <name>.context one row per table position: down, up, flag for A, C, G, T
  * generateCountList + storeCountList (VC:387-394, 580-627) on the committed tumour directory
                                                    -> toy_subset_TUMOUR.order
The tables are the reference-written ones already committed under tests/golden/ plus `context_edge.txt`, a table written
HERE (data: homopolymer runs around the 18-of-21 decision, gaps at every offset, soft-masked / N / '.' reference cells,
repeated rows whose later copies differ, CRLF line ends, threshold cells -1_-1 / 0_0 / exponents).
Outputs are data only (the reference's outputs); big ones are gzipped with a fixed mtime.
"""
import gzip
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
DRV = os.path.join(ROOT, "oracle", "_ref", "vc_ref_driver")
OUT = os.path.join(HERE, "vc_ref")

TABLES = {
    "toy_subset_0.0020": "toy_subset/expected_positionSpecificNoise_0.0020.txt",
    "toy_subset_0.0100_cov500": "toy_subset/expected_positionSpecificNoise_0.0100_cov500.txt",
    "toy_subset_default_0.0120": "toy_subset/expected_positionSpecificNoise_default_0.0120.txt",
    "mini_edge_0.0020_cov100": "mini_edge/expected_positionSpecificNoise_0.0020_cov100.txt",
    "mini_edge_0.0005_cov1": "mini_edge/expected_positionSpecificNoise_0.0005_cov1.txt",
    "mini_edge_0.0500_cov1000": "mini_edge/expected_positionSpecificNoise_0.0500_cov1000.txt",
    "mini_edge_default_7": "mini_edge/expected_positionSpecificNoise_default_7.txt",
    "irregular_0.0020_cov100": "irregular/expected_positionSpecificNoise_0.0020_cov100.txt",
    "context_edge": "vc_ref/context_edge.txt",
    "toy_full_0.0020": "toy/positionSpecificNoise_0.0020.txt.gz",  # Toy_data, all 41 486 rows
}
HEADER = "chrom\tposition\treference\tduplicate\tThres_A\tThres_C\tThres_G\tThres_T\tGerm_Max_A\tGerm_Max_C\tGerm_Max_G\tGerm_Max_T"


def write_context_edge(path):
    rng = np.random.default_rng(20261004)
    rows = []

    def row(c, p, ref, dup="NO", thr=None, germ=None):
        thr = thr or ["0.002000_0.002000"] * 4
        germ = germ or ["0", "0", "0", "-"]
        if ref in "ACGT":
            thr = list(thr)
            thr["ACGT".index(ref)] = "-2_-2"
        rows.append("\t".join([c, str(p), ref, dup] + list(thr) + list(germ)))

    # chrH1: a long A run with single other bases dropped in: the A+X pair count walks across 18 / 19
    seq = ["A"] * 70
    for i in (5, 17, 26, 29, 41, 42, 55, 63):
        seq[i] = "CGT"[i % 3]
    for i, b in enumerate(seq):
        row("chrH1", 100 + i, b)
    # chrH2: AC dinucleotide repeat with G/T interruptions of growing density
    for i in range(80):
        b = "AC"[i & 1]
        if i > 20 and rng.random() < (i - 20) / 120:
            b = "GT"[int(rng.integers(0, 2))]
        row("chrH2", 5000 + i, b)
    # chrH3: gaps -- every offset -10..+10 is missing for some position (islands of 1-12 positions, holes of 1-4)
    p = 10
    while p < 400:
        n = int(rng.integers(1, 13))
        for i in range(n):
            row("chrH3", p + i, "ACGT"[int(rng.integers(0, 4))] if rng.random() < 0.5 else "T")
        p += n + int(rng.integers(1, 5))
    # chrH4: reference cells that are not upper-case A/C/G/T (case is preserved)
    for i in range(60):
        row("chrH4", 900 + i, ["a", "c", "g", "t", "N", "n", ".", "A", "C", "G", "T", "T", "T"][int(rng.integers(0, 13))])
    # chrH5: repeated rows (the first wins, VC:505 insert), duplicate = YES, unusual threshold / germ-max cells
    for i in range(40):
        b = "ACGT"[int(rng.integers(0, 4))]
        thr = [["0.002000_0.002000", "-1_-1", "0_0", "0.01_0.01", "1e-3_2.5E-4", "0.000001_0.999999", "0_0.05"][int(rng.integers(0, 7))] for _ in range(4)]
        germ = [["-", "0", "-888", "0.00170648", "0.0499079", "1e-05"][int(rng.integers(0, 6))] for _ in range(4)]
        row("chrH5", 70 + i, b, "YES" if i % 7 < 2 else "NO", thr, germ)
        if i % 7 == 0:  # the same position again, different in every cell: must be ignored
            row("chrH5", 70 + i, "ACGT"[("ACGT".index(b) + 1) % 4], "NO", ["0.5_0.5"] * 4, ["0.04"] * 4)
        if i % 7 == 1:  # and an exact copy, as the writer produces for overlapping amplicons
            rows.append(rows[-1])
    # the last chromosome name is a prefix of another one: keys are chrom_pos strings
    for i in range(12):
        row("chr1", 1 + i, "G")
        row("chr11", 1 + i, "C")
    with open(path, "w", newline="") as f:
        f.write(HEADER + "\r\n" + "\r\n".join(rows) + "\r\n")  # CRLF: "%s" strips the \r (VC:476)
    return len(rows)


def sha(b):
    return hashlib.sha256(b).hexdigest()


def put(name, data, gz):
    path = os.path.join(OUT, name + (".gz" if gz else ""))
    if gz:
        with gzip.GzipFile(path, "wb", mtime=0) as f:
            f.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)


def main():
    assert os.path.exists(DRV), "run `make -C oracle` first"
    os.makedirs(OUT, exist_ok=True)
    n = write_context_edge(os.path.join(OUT, "context_edge.txt"))
    print("context_edge.txt:", n, "rows")
    tmp = tempfile.mkdtemp(prefix="ampli_golden_vc_")
    digests = []
    for name, rel in TABLES.items():
        src = os.path.join(HERE, rel)
        if src.endswith(".gz"):
            table = os.path.join(tmp, name + ".txt")
            open(table, "wb").write(gzip.open(src).read())
        else:
            table = src
        vcf = os.path.join(tmp, name + ".vcf")
        big = name.startswith("toy_full")
        for mode in ("maps", "context"):
            r = subprocess.run([DRV, mode, table, vcf], capture_output=True)
            assert r.returncode == 0, r.stderr.decode()
            digests.append(f"{name}.{mode} sha256 {sha(r.stdout)} lines {len(r.stdout.splitlines())}")
            if not (big and mode == "maps"):  # the 374 k map entries of the full toy table travel as a digest only
                put(f"{name}.{mode}", r.stdout, gz=True)
        v = open(vcf, "rb").read()
        digests.append(f"{name}.vcf sha256 {sha(v)} lines {len(v.splitlines())}")
        print(name, "ok")
    # the tumour directory's visit order (literal path: the order is a function of the hash of the listed strings)
    lit = "/root/repo/tests/golden/toy_subset/TUMOUR"
    assert os.path.realpath(lit) == os.path.realpath(os.path.join(HERE, "toy_subset", "TUMOUR")), "run from /root/repo"
    r = subprocess.run([DRV, "order", lit, os.path.join(tmp, "list.txt")], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    put("toy_subset_TUMOUR.order", r.stdout, gz=False)
    with open(os.path.join(OUT, "digests.txt"), "w") as f:
        f.write("\n".join(digests) + "\n")
    print(open(os.path.join(OUT, "digests.txt")).read())


if __name__ == "__main__":
    sys.exit(main())
