"""Cohorts outside the exactness envelope of the threshold sums (a coverage cut-off of a few reads with depths in the tens of
millions).  Inside the envelope every partial sum of EE:1597 is exact and any order of addition gives the reference's double; outside it
the double depends on the order -- and the reference still writes a table.  Round 6: instead of refusing, the sums are formed once more in
the reference's own order (ampli_error_sums_inorder).  Pinned here:
  * the ORDER: estimateThresholds walks `equal_range` of the reference's multimap (EE:1555, 1565); the reference's own container, in the
    reference's own build (oracle/_ref/ee_ref_driver ... dump -> .walk), hands a key's records out in reverse insertion order -- last file
    of the visit order first, a position's later lines of a file before its first -- which is what the oracle and the kernel walk (CPU);
  * the ARITHMETIC: oracle_error_sums_inorder restates `sum = sum + X + float(RD)*float(C)` in that order; its table equals the
    reference's on such cohorts (CPU), the kernel's sums equal the oracle's bit for bit, chunked or not (GPU), and the command line writes
    the reference's table instead of refusing (GPU)."""
import os
import subprocess

import numpy as np
import pytest

from amplisolve_amd.hostio import HostCohort
from oracle import pyoracle as orc
from tests.helpers import write_envelope_panel, write_fresh_panel

need_ref = pytest.mark.skipif(not os.path.exists(orc.REF_EE_DRIVER), reason="oracle/_ref/ee_ref_driver is absent (make -C oracle where /root/reference exists)")
ABSENT = np.iinfo(np.int32).min
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "amplisolve_amd", "bin")


def _reference(d, C, cov, dump=False):
    (d / "o").mkdir(exist_ok=True)
    r = subprocess.run([orc.REF_EE_DRIVER, "p.bed", "r.txt", "d.txt", "N", C, cov, "o"] + (["dump"] if dump else []), capture_output=True, text=True, cwd=d)
    assert r.returncode == 0, r.stderr[-400:]
    name = [n for n in os.listdir(d / "o") if n.startswith("positionSpecificNoise_")]
    assert len(name) == 1
    return (d / "o" / name[0]).read_bytes(), name[0]


@need_ref
@pytest.mark.parametrize("kind,seed,S", [("envelope", 1, 9), ("envelope", 2, 70), ("fresh_triple", 3, 12)])
def test_the_references_own_walk_is_reverse_insertion_order(tmp_path, monkeypatch, kind, seed, S):
    d = tmp_path
    if kind == "envelope":
        write_envelope_panel(d, seed, S=S)
    else:
        from tests.test_panel_variants_vs_reference import _vary

        write_fresh_panel(d, seed, S=S, amplicons=4)
        _vary(d, np.random.default_rng(seed), ("aseq_triple", "aseq_shuffled", "aseq_header_only"))
    _reference(d, "0.002", "1", dump=True)
    monkeypatch.chdir(d)
    co = HostCohort("p.bed", "N", refbases_file="r.txt")
    walk = {l.split("\t")[0]: l.rstrip("\n").split("\t")[1:] for l in open("dump.walk")}
    n = 0
    for p in range(co.P):
        c, x = co.position(p)
        slots = [co.P + e for e in range(int(co.dup_off[p + 1]) - 1, int(co.dup_off[p]) - 1, -1)] + [p]  # later lines first, then the first one
        for nt in range(4):
            exp = []
            for s in range(co.S - 1, -1, -1):  # the last file of the visit order first
                for r_ in slots:
                    rec = co.recs[s, r_].astype(np.int64)
                    if rec[0] != ABSENT:
                        exp.append(f"{rec[nt]}_{rec[:4].sum()}_{rec[4 + nt]}_{rec[4:].sum()}")  # "Xfw_FW_Xbw_BW", EE:1236-1245
            assert walk[f"{c}_{x}_{'ACGT'[nt]}"] == exp, (c, x, nt)
            n += 1
    assert n == 4 * co.P and co.E > 0


@need_ref
@pytest.mark.parametrize("seed,C,cov,S", [(1, "0.002", "1", 9), (2, "0.0005", "1", 14), (3, "0.03", "2", 30)])
def test_oracle_inorder_table_equals_the_references(tmp_path, monkeypatch, seed, C, cov, S):
    d = tmp_path
    write_envelope_panel(d, seed, S=S)
    want, _ = _reference(d, C, cov)
    monkeypatch.chdir(d)
    co = HostCohort("p.bed", "N", refbases_file="r.txt")
    acc = orc.error_reduce(co.recs, co.P, float(C), int(cov), E=co.E, dup_off=co.dup_off)
    assert acc["order_sensitive"] == 1  # the cohort IS outside the envelope: forward and backward sums differ in some double
    snt = orc.error_sums_inorder(co.recs, co.P, float(C), int(cov), E=co.E, dup_off=co.dup_off)
    assert (snt != acc["snt"]).sum() > 10
    assert np.allclose(snt, acc["snt"], rtol=1e-12)
    acc["snt"] = snt
    fin = orc.error_finalize(acc)
    co.write_error_table(fin["rate"], fin["code"], fin["germ_val"].astype(np.float32), fin["germ_present"], "ours.txt")
    assert open("ours.txt", "rb").read() == want


@pytest.mark.gpu
@pytest.mark.parametrize("layout,deep", [("i32", 45_000_000), ("u24", 4_000_000), ("u16", 30_000)])
@pytest.mark.parametrize("cuts", [(0, 13), (0, 1, 6, 13), (0, 12, 13)])
def test_kernel_sums_equal_the_oracles_bit_for_bit(ctx, layout, deep, cuts):
    """ampli_error_sums_inorder against oracle_error_sums_inorder: one chunk, or chunks walked from the last to the first with the sums
    carried in the table; positions listed twice; every record layout (depths as deep as the layout holds)."""
    from tests.test_gpu_parity import _t
    from tests.test_gpu_records import _pack

    rng = np.random.default_rng(deep % 1000 + len(cuts))
    P, S = 333, 13
    mult = np.zeros(P, np.int64)
    mult[rng.choice(P, 40, replace=False)] = 1
    mult[rng.choice(P, 6, replace=False)] = 2
    dup_off = np.concatenate([[0], np.cumsum(mult)]).astype(np.uint32)
    E = int(dup_off[-1])
    R = P + E
    recs = np.zeros((S, R, 8), np.int32)
    few = rng.random((S, R)) < 0.35
    fw = np.where(few, rng.integers(1, 9, (S, R)), rng.integers(deep // 2, deep, (S, R)))
    bw = np.where(few, rng.integers(1, 9, (S, R)), rng.integers(deep // 2, deep, (S, R)))
    for st, depth in ((0, fw), (4, bw)):
        alts = (depth[:, :, None] * rng.uniform(0, 0.0166, (S, R, 3))).astype(np.int64) * (~few)[:, :, None]
        recs[:, :, st + 1: st + 4] = alts
        recs[:, :, st] = depth - alts.sum(-1)
    recs[rng.random((S, R)) < 0.07] = np.array([ABSENT, 0, 0, 0, 0, 0, 0, 0], np.int32)
    C_value, cov = 0.002, 1
    want = orc.error_sums_inorder(recs, P, C_value, cov, E=E, dup_off=dup_off)
    fwd = orc.error_reduce(recs, P, C_value, cov, E=E, dup_off=dup_off)
    if layout == "i32":
        assert fwd["order_sensitive"] == 1
    acc = ctx.new_acc(P)
    acc.buf.fill_(0x5A)
    chunks = []
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        prim = np.ascontiguousarray(recs[lo:hi, :P])
        ext = np.ascontiguousarray(recs[lo:hi, P:])
        chunks.append(ctx.records(_pack(ctx, prim, layout), layout, hi - lo, E=E, ext=_pack(ctx, ext, layout), dup_off=_t(dup_off)))
    for k in range(len(chunks) - 1, -1, -1):
        ctx.error_sums_inorder(chunks[k], P, acc, C_value, cov, accumulate=k != len(chunks) - 1)
    got = acc.snt.cpu().numpy()
    assert np.array_equal(got.view(np.int64), want.view(np.int64))
    assert ctx.flags() == 0


@need_ref
@pytest.mark.gpu
@pytest.mark.parametrize("seed,C,cov,S,chunk,shape", [(1, "0.002", "1", 9, None, ()), (2, "0.0005", "1", 14, "1", ()), (3, "0.03", "2", 30, "40000", ()),
                                                      (4, "0.002", "1", 12, "30000", ("aseq_own_rd", "aseq_triple", "aseq_shuffled", "aseq_header_only"))])
def test_command_line_writes_the_references_table_instead_of_refusing(tmp_path, seed, C, cov, S, chunk, shape):
    """... also with lines that carry their own RD column (the RD plane travels with every resident chunk of the in-order pass), a position
    listed three times, shuffled lines and a header-only file."""
    d = tmp_path
    write_envelope_panel(d, seed, S=S)
    if shape:
        from tests.test_panel_variants_vs_reference import _vary

        _vary(d, np.random.default_rng(seed), shape)
    want, name = _reference(d, C, cov)
    env = dict(os.environ, AMPLISOLVE_STRICT_EXIT="1", AMPLISOLVE_REFBASES_FILE="r.txt")
    if chunk:
        env["AMPLISOLVE_CHUNK_BYTES"] = chunk
    r = subprocess.run([f"{BIN}/AmpliSolveErrorEstimation", "panel_design=p.bed", "reference_genome=x.fa", "germline_dir=N", f"C_value={C}",
                        f"coverage_cutoff={cov}", "default_error=0.01", "output_dir=q"], capture_output=True, text=True, cwd=d, env=env)
    assert r.returncode == 0, r.stdout[-600:] + r.stderr[-300:]
    assert "summing again in the reference's order" in r.stdout
    assert (d / "q" / name).read_bytes() == want
