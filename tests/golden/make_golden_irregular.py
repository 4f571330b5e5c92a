#!/usr/bin/env python3
"""Golden fixture for lines whose RD column is not A+C+G+T (EE:1178-1181: "malakia paizei edo", then the line is used with
its own RD: Germ_Max AF = X / RD, EE:1229-1232) -- generated from the REFERENCE ITSELF (oracle/_ref/ee_ref_driver =
AmpliSolveErrorEstimation.cpp compiled where it lies).  Build container only:   python tests/golden/make_golden_irregular.py
Outputs are data only: the inputs and the reference's tables under tests/golden/irregular/."""
import os
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import DRV, NT, read_bed_walk, run_ref, write_ref_and_dups  # noqa: E402


def main():
    assert os.path.exists(DRV), "run `make -C oracle` first"
    out = os.path.join(HERE, "irregular")
    shutil.rmtree(out, ignore_errors=True)
    os.makedirs(os.path.join(out, "NORMAL"))
    rng = np.random.default_rng(20261004)
    bed_rows = [("chr3", 200, 289), ("chr3", 270, 319), ("chrX", 10, 59)]  # the first two overlap: positions listed twice
    with open(os.path.join(out, "panel.bed"), "w") as f:
        for i, (c, a, b) in enumerate(bed_rows):
            f.write(f"{c}\t{a}\t{b}\tAMPL{i}\trs{i}\tGENE{i}\n")
    walk = read_bed_walk(os.path.join(out, "panel.bed"))
    base = {k: NT[rng.integers(0, 4)] for k in set(walk)}
    write_ref_and_dups(walk, base, os.path.join(out, "refbases.txt"), os.path.join(out, "dups.txt"))
    S, n_irr = 9, 0
    for s in range(S):
        with open(os.path.join(out, "NORMAL", f"R{s}.PILEUP.ASEQ"), "w") as f:
            f.write("chr\tpos\tdbsnp\tMAF\tref\talt\tA\tC\tG\tT\tRD\tArs\tCrs\tGrs\tTrs\n")
            for (c, p) in walk:
                if rng.random() < 0.08:
                    continue
                refnt = NT.index(base[(c, p)])
                depth = [int(rng.choice([90, 100, 150, 400, 1200, 5000])) for _ in range(2)]
                cnts = np.zeros((2, 4), np.int64)
                for st in range(2):
                    left = depth[st]
                    for nt in range(4):
                        if nt == refnt:
                            continue
                        fr = rng.choice([0, 0, 0.001, 0.004, 0.02, 0.04, 0.049, 0.06, 0.09])
                        k = min(left, int(depth[st] * fr) + int(rng.integers(0, 2)))
                        cnts[st, nt] = k
                        left -= k
                    cnts[st, refnt] = left
                tot = cnts.sum(0)
                rd = int(tot.sum())
                u = rng.random()
                if u < 0.25:  # every fourth line is irregular, in ways that move the AF <= 0.05 gate both ways
                    rd = int(rng.choice([rd * 2, rd + 37, max(1, rd // 2), max(1, rd // 3), rd - 1, 0, 7, rd * 40]))
                    n_irr += rd != int(tot.sum())
                f.write(f"{c}\t{p}\t.\t.\t.\t.\t{tot[0]}\t{tot[1]}\t{tot[2]}\t{tot[3]}\t{rd}\t{cnts[1,0]}\t{cnts[1,1]}\t{cnts[1,2]}\t{cnts[1,3]}\n")
    lit = "/root/repo/tests/golden/irregular/NORMAL"
    tmp = tempfile.mkdtemp(prefix="ampli_irr_")
    for C_value, cov, tag in (("0.002", "100", "0.0020_cov100"), ("0.01", "1", "0.0100_cov1")):
        o = os.path.join(tmp, tag)
        r = run_ref(os.path.join(out, "panel.bed"), os.path.join(out, "refbases.txt"), os.path.join(out, "dups.txt"), lit, C_value, cov, o,
                    os.path.join(tmp, tag + "_dump"))
        assert r.stdout.count("malakia paizei edo") == n_irr, (r.stdout.count("malakia paizei edo"), n_irr)
        src = [f for f in os.listdir(o) if f.startswith("positionSpecificNoise_")][0]
        shutil.copy(os.path.join(o, src), os.path.join(out, f"expected_positionSpecificNoise_{tag}.txt"))
    shutil.copy(os.path.join(tmp, "0.0020_cov100_dump.order"), os.path.join(out, "expected_visit_order.txt"))
    shutil.rmtree(tmp, ignore_errors=True)
    print("irregular:", len(walk), "walk positions,", S, "samples,", n_irr, "irregular lines")


if __name__ == "__main__":
    main()
