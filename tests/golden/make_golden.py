#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the REFERENCE ITSELF.

Run in the build container only (needs /root/reference and oracle/_ref, i.e. `make -C oracle`):
    python tests/golden/make_golden.py

What it runs (never copying reference sources):
  * oracle/_ref/ee_ref_driver  = AmpliSolveErrorEstimation.cpp compiled from /root/reference, its own
    storeReference/storeDuplicates/storeCountList/storeGermlineStatistics/estimateThresholds/
    generateFinalOutput called in main()'s order (EE:426-454) on (a) Toy_data, (b) a subset of Toy_data
    small enough to commit as a fixture, (c) a synthetic edge-case mini panel.
  * oracle/_ref/libvc_scorer_ref.so = the Boost-free Poisson scorer lines of AmpliSolveVariantCalling.cpp
    (VC:3720-3795, 3816-3884) on a (k, RD, err) grid.
Outputs are data only: inputs (BED / ASEQ / ref-base tables) and the reference's outputs.
"""
import ctypes as C
import gzip
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"
TOY = f"{REF}/Toy_data"
DRV = os.path.join(ROOT, "oracle", "_ref", "ee_ref_driver")
SCORER = os.path.join(ROOT, "oracle", "_ref", "libvc_scorer_ref.so")
NT = "ACGT"


def sha256(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def read_bed_walk(bed):
    walk = []
    for line in open(bed):
        f = line.split()
        if len(f) >= 3:
            walk += [(f[0], i) for i in range(int(f[1]), int(f[2]) + 1)]
    return walk


def derive_ref_table(aseq_files):
    """The toy ASEQ `ref` column is '.', so the panel reference bases are DERIVED: summed A/C/G/T over all
    files, first strictly-largest base; never-seen positions -> N."""
    tot = {}
    for fn in aseq_files:
        with open(fn) as f:
            next(f)
            for line in f:
                x = line.split()
                if len(x) < 15:
                    continue
                k = (x[0], int(x[1]))
                v = np.array([int(x[6]), int(x[7]), int(x[8]), int(x[9])])
                tot[k] = tot.get(k, 0) + v
    return {k: NT[int(np.argmax(v))] if v.max() > 0 else "N" for k, v in tot.items()}


def write_ref_and_dups(walk, base_of, ref_path, dup_path):
    seen, dups = set(), []
    with open(ref_path, "w") as f:
        for c, p in walk:
            f.write(f"{c}\t{p}\t{base_of.get((c, p), 'N')}\n")
            if (c, p) in seen and (c, p) not in dups:
                dups.append((c, p))
            seen.add((c, p))
    with open(dup_path, "w") as f:
        for c, p in sorted(dups):
            f.write(f"{c}\t{p}\n")


def run_ref(bed, refb, dups, gdir, C_value, cov, out_dir, dump):
    os.makedirs(out_dir, exist_ok=True)
    r = subprocess.run([DRV, bed, refb, dups, gdir, C_value, cov, out_dir, dump], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r


def gz_copy(src, dst):
    with open(src, "rb") as a, gzip.GzipFile(dst, "wb", mtime=0) as b:
        shutil.copyfileobj(a, b)


def main():
    assert os.path.exists(DRV) and os.path.exists(SCORER), "run `make -C oracle` first"
    tmp = tempfile.mkdtemp(prefix="ampli_golden_")

    # ---------------- (a) full Toy_data: digest + the complete table, gzipped ----------------
    bed = f"{TOY}/AmpliSeq_30genes_Designed-1.bed"
    normals = sorted(os.path.join(f"{TOY}/NORMAL_ASEQ_DIR", f) for f in os.listdir(f"{TOY}/NORMAL_ASEQ_DIR") if f.endswith(".ASEQ"))
    tumours = sorted(os.path.join(f"{TOY}/TUMOUR_ASEQ_DIR", f) for f in os.listdir(f"{TOY}/TUMOUR_ASEQ_DIR") if f.endswith(".ASEQ"))
    base_of = derive_ref_table(normals + tumours)
    walk = read_bed_walk(bed)
    d = os.path.join(HERE, "toy")
    os.makedirs(d, exist_ok=True)
    refb, dups = os.path.join(tmp, "toy_refbases.txt"), os.path.join(tmp, "toy_dups.txt")
    write_ref_and_dups(walk, base_of, refb, dups)
    run_ref(bed, refb, dups, f"{TOY}/NORMAL_ASEQ_DIR", "0.002", "100", os.path.join(tmp, "toy_out"), os.path.join(tmp, "toy"))
    table = os.path.join(tmp, "toy_out", "positionSpecificNoise_0.0020.txt")
    gz_copy(table, os.path.join(d, "positionSpecificNoise_0.0020.txt.gz"))
    gz_copy(refb, os.path.join(d, "refbases.txt.gz"))
    shutil.copy(os.path.join(tmp, "toy.order"), os.path.join(d, "visit_order.txt"))
    with open(os.path.join(d, "digests.txt"), "w") as f:
        f.write(f"positionSpecificNoise_0.0020.txt sha256 {sha256(table)}\n")
        f.write("# germline_dir literal: /root/reference/Toy_data/NORMAL_ASEQ_DIR ; C=0.002 cov=100 ; SURVEY App. D digest ebb19204...826ba3a\n")
    print("toy table sha256", sha256(table))

    # ---------------- (b) Toy_data subset small enough to travel ----------------
    sub = os.path.join(HERE, "toy_subset")
    shutil.rmtree(sub, ignore_errors=True)
    os.makedirs(os.path.join(sub, "NORMAL"))
    os.makedirs(os.path.join(sub, "TUMOUR"))
    rows = [l for l in open(bed).read().splitlines() if l.strip()]
    # amplicons that overlap another one (duplicated positions), chrX, and the first few on chr8
    cnt = {}
    for k in walk:
        cnt[k] = cnt.get(k, 0) + 1
    keep = []
    for i, l in enumerate(rows):
        f = l.split()
        span = [(f[0], x) for x in range(int(f[1]), int(f[2]) + 1)]
        has_dup = any(cnt[k] > 1 for k in span)
        if i < 3 or (has_dup and len([r for r in keep if r.split()[0] == f[0]]) < 3) or (f[0] == "chrX" and sum(r.startswith("chrX") for r in keep) < 2):
            keep.append(l)
    with open(os.path.join(sub, "panel.bed"), "w", newline="") as f:
        f.write("\r\n".join(keep) + "\r\n")  # the toy BED has CRLF line ends
    swalk = read_bed_walk(os.path.join(sub, "panel.bed"))
    sset = set(swalk)
    for files, name in ((normals, "NORMAL"), (tumours, "TUMOUR")):
        for fn in files:
            with open(fn) as src, open(os.path.join(sub, name, os.path.basename(fn)), "w") as dst:
                dst.write(next(src))
                for line in src:
                    x = line.split("\t", 2)
                    if (x[0], int(x[1])) in sset:
                        dst.write(line)
    write_ref_and_dups(swalk, base_of, os.path.join(sub, "refbases.txt"), os.path.join(sub, "dups.txt"))
    # literal directory string the tests will use too: the visit order hashes it
    lit = "/root/repo/tests/golden/toy_subset/NORMAL"
    assert os.path.realpath(lit) == os.path.realpath(os.path.join(sub, "NORMAL")), "run from /root/repo"
    run_ref(os.path.join(sub, "panel.bed"), os.path.join(sub, "refbases.txt"), os.path.join(sub, "dups.txt"), lit, "0.002", "100",
            os.path.join(tmp, "sub_out"), os.path.join(tmp, "sub"))
    shutil.copy(os.path.join(tmp, "sub_out", "positionSpecificNoise_0.0020.txt"), os.path.join(sub, "expected_positionSpecificNoise_0.0020.txt"))
    shutil.copy(os.path.join(tmp, "sub.order"), os.path.join(sub, "expected_visit_order.txt"))
    with open(os.path.join(tmp, "sub.counts")) as f, open(os.path.join(sub, "expected_counts.txt"), "w") as g:
        g.write("".join(sorted(f.readlines())))
    # a second parameter set on the same inputs
    run_ref(os.path.join(sub, "panel.bed"), os.path.join(sub, "refbases.txt"), os.path.join(sub, "dups.txt"), lit, "0.01", "500",
            os.path.join(tmp, "sub_out2"), os.path.join(tmp, "sub2"))
    shutil.copy(os.path.join(tmp, "sub_out2", "positionSpecificNoise_0.0100.txt"), os.path.join(sub, "expected_positionSpecificNoise_0.0100_cov500.txt"))
    print("toy subset:", len(swalk), "walk positions,", len(sset), "unique")

    # ---------------- (c) synthetic edge-case mini panel ----------------
    mini = os.path.join(HERE, "mini_edge")
    shutil.rmtree(mini, ignore_errors=True)
    os.makedirs(os.path.join(mini, "NORMAL"))
    rng = np.random.default_rng(20240607)
    bed_rows = [("chr1", 1000, 1099), ("chr1", 1080, 1129), ("chr2", 500, 579), ("chrX", 70, 129), ("chr1", 1120, 1140), ("chrM", 5, 24)]
    with open(os.path.join(mini, "panel.bed"), "w") as f:
        for i, (c, a, b) in enumerate(bed_rows):
            f.write(f"{c}\t{a}\t{b}\tAMPL{i}\trs{i}\tGENE{i}\n")
    mwalk = read_bed_walk(os.path.join(mini, "panel.bed"))
    mbase = {k: NT[rng.integers(0, 4)] for k in set(mwalk)}
    for k in list(mbase)[::17]:
        mbase[k] = "N"
    for k in list(mbase)[5::23]:
        mbase[k] = mbase[k].lower()  # soft-masked base: neither -2_-2 nor any call
    write_ref_and_dups(mwalk, mbase, os.path.join(mini, "refbases.txt"), os.path.join(mini, "dups.txt"))
    S = 13
    for s in range(S):
        with open(os.path.join(mini, "NORMAL", f"M{s:02d}_x.PILEUP.ASEQ"), "w") as f:
            f.write("chr\tpos\tdbsnp\tMAF\tref\talt\tA\tC\tG\tT\tRD\tArs\tCrs\tGrs\tTrs\n")
            for (c, p) in mwalk:  # duplicated walk positions appear twice, like real ASEQ output
                if rng.random() < 0.12:
                    continue  # absent line
                refnt = NT.index(mbase[(c, p)].upper()) if mbase[(c, p)].upper() in NT else 0
                depth = [int(rng.choice([0, 1, 60, 99, 100, 101, 150, 800, 3000, 33000])) for _ in range(2)]
                cnts = np.zeros((2, 4), np.int64)
                for st in range(2):
                    left = depth[st]
                    for nt in range(4):
                        if nt == refnt:
                            continue
                        fr = rng.choice([0, 0, 0, 0.0004, 0.002, 0.01, 0.049, 0.05, 0.051, 0.3, 0.5])
                        k = min(left, int(depth[st] * fr) + int(rng.integers(0, 2)))
                        cnts[st, nt] = k
                        left -= k
                    cnts[st, refnt] = left
                tot = cnts.sum(0)
                f.write(f"{c}\t{p}\t.\t.\t.\t.\t{tot[0]}\t{tot[1]}\t{tot[2]}\t{tot[3]}\t{tot.sum()}\t{cnts[1,0]}\t{cnts[1,1]}\t{cnts[1,2]}\t{cnts[1,3]}\n")
    lit = "/root/repo/tests/golden/mini_edge/NORMAL"
    for C_value, cov, tag in (("0.002", "100", "0.0020_cov100"), ("0.0005", "1", "0.0005_cov1"), ("0.05", "1000", "0.0500_cov1000")):
        out = os.path.join(tmp, f"mini_{tag}")
        run_ref(os.path.join(mini, "panel.bed"), os.path.join(mini, "refbases.txt"), os.path.join(mini, "dups.txt"), lit, C_value, cov, out,
                os.path.join(tmp, f"mini_{tag}_dump"))
        src = [f for f in os.listdir(out) if f.startswith("positionSpecificNoise_")][0]
        shutil.copy(os.path.join(out, src), os.path.join(mini, f"expected_positionSpecificNoise_{tag}.txt"))
        with open(os.path.join(tmp, f"mini_{tag}_dump.counts")) as f, open(os.path.join(mini, f"expected_counts_{tag}.txt"), "w") as g:
            g.write("".join(sorted(f.readlines())))
    shutil.copy(os.path.join(tmp, "mini_0.0020_cov100_dump.order"), os.path.join(mini, "expected_visit_order.txt"))
    print("mini_edge:", len(mwalk), "walk positions,", S, "samples")

    # ---------------- (d) Poisson scorer grid from the reference's own functions ----------------
    L = C.CDLL(SCORER)
    L.ref_kf_gammaq.restype = C.c_double
    L.ref_kf_gammaq.argtypes = [C.c_double, C.c_double]
    L.ref_kf_lgamma.restype = C.c_double
    L.ref_kf_lgamma.argtypes = [C.c_double]
    ks = np.array([0, 1, 2, 3, 4, 5, 6, 8, 10, 13, 17, 20, 25, 32, 40, 50, 64, 80, 99, 100, 101, 150, 250, 500, 1000, 2500, 5000, 12000], np.int32)
    rds = np.array([0, 1, 50, 99, 100, 101, 250, 500, 997, 1000, 2500, 5000, 12500, 25000, 33395, 50000], np.int32)
    errs = np.array([-1.0, 0.0, 0.0001, 0.0005, 0.001, 0.002, 0.002189, 0.0035, 0.01, 0.02, 0.05, 0.25], np.float32)
    K, R, E = np.meshgrid(ks, rds, errs, indexing="ij")
    k, rd, err = K.ravel().astype(np.int32), R.ravel().astype(np.int32), E.ravel().astype(np.float32)
    rngk = np.random.default_rng(99)
    n2 = 6000  # plus random points near the call boundary (k a little above the mean)
    rd2 = rngk.integers(100, 40000, n2).astype(np.int32)
    err2 = rngk.choice(errs[2:], n2).astype(np.float32)
    k2 = np.maximum(0, (rd2 * err2.astype(np.float64) + rngk.normal(0, 1, n2) * np.sqrt(rd2 * err2.astype(np.float64) + 1) * 2).astype(np.int32))
    k, rd, err = np.concatenate([k, k2]), np.concatenate([rd, rd2]), np.concatenate([err, err2])
    q = np.empty(k.size, np.float64)
    L.ref_score_batch(k.ctypes.data_as(C.c_void_p), rd.ctypes.data_as(C.c_void_p), err.ctypes.data_as(C.c_void_p), C.c_long(k.size), q.ctypes.data_as(C.c_void_p))
    s_ = rngk.uniform(0.5, 3000, 4000)
    z_ = s_ * rngk.uniform(0.05, 3.0, 4000)
    gq = np.array([L.ref_kf_gammaq(float(a), float(b)) for a, b in zip(s_, z_)])
    lg = np.array([L.ref_kf_lgamma(float(a)) for a in s_])
    np.savez_compressed(os.path.join(HERE, "vc_scorer_reference.npz"), k=k, rd=rd, err=err, q=q, s=s_, z=z_, gammaq=gq, lgamma=lg)
    print("scorer grid:", k.size, "score points,", s_.size, "gammaq points")
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
