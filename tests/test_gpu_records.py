"""The explicit cohort description (ampli_records) the streaming command lines use: chunks of samples reduced one
launch at a time into the same table, padded row strides, extras in their own array -- all against the oracle's single
pass over the dense array; and the launch-shape knobs of poisson_call, which must never change a result."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tests.helpers import edge_case_recs, synth_recs, synth_ref
from tests.test_gpu_parity import _t, assert_acc_equal, assert_final_equal

pytestmark = pytest.mark.gpu
ABSENT = np.iinfo(np.int32).min


def _pack(ctx, recs32, layout):
    """host int32 [n][stride][8] -> device tensor in `layout`"""
    t = _t(recs32)
    if layout == "i32":
        return t
    out, fits = ctx.pack(t, layout)
    assert fits
    return out


def _panel_with_extras(P, S, rng):
    mult = np.zeros(P, np.int64)
    mult[rng.choice(P, max(1, P // 8), replace=False)] = 1
    mult[rng.choice(P, max(1, P // 50), replace=False)] = 2
    dup_off = np.concatenate([[0], np.cumsum(mult)]).astype(np.uint32)
    E = int(dup_off[-1])
    recs = np.concatenate([synth_recs(P, S), edge_case_recs(E, S, rng)], axis=1)
    return recs, E, dup_off


@pytest.mark.parametrize("layout", ["i32", "u24", "u16"])
@pytest.mark.parametrize("P,S,cuts,splits", [(300, 13, (0, 13), 0), (300, 13, (0, 1, 2, 13), 0), (1000, 40, (0, 7, 8, 29, 40), 0),
                                            (65, 50, (0, 25, 50), 2), (4097, 37, (0, 16, 32, 37), 0)])
def test_chunks_accumulate_to_the_single_pass(ctx, layout, P, S, cuts, splits):
    """A cohort reduced chunk by chunk (each chunk its own buffers, padded row stride, extras apart, and its own slot
    layout for the extras it actually holds) equals the oracle's single pass; the last launch finalises as well."""
    rng = np.random.default_rng(P + S)
    recs, E, dup_off = _panel_with_extras(P, S, rng)
    if layout == "u16":
        recs = np.where(recs == ABSENT, ABSENT, np.minimum(recs, 65534)).astype(np.int32)
    ref = orc.error_reduce(recs, P, 0.002, 100, E=E, dup_off=dup_off)
    ref_fin = orc.error_finalize(ref)
    ctx.set_tuning(splits)
    acc = ctx.new_acc(P)
    acc.buf.fill_(0x5A)  # the first chunk must overwrite, not accumulate into, whatever was there
    fin = None
    for ci in range(len(cuts) - 1):
        lo, hi = cuts[ci], cuts[ci + 1]
        n = hi - lo
        chunk = recs[lo:hi]
        # this chunk's own extras layout: per position only as many slots as some sample of the chunk fills (slots are
        # visited in order, so only the unused TAIL of a position's run may go); every other chunk keeps them all
        used = (chunk[:, P:, 0] != ABSENT).any(axis=0) if E else np.zeros(0, bool)
        keep = np.zeros(E, bool)
        mult_c = np.zeros(P, np.int64)
        for p in range(P):
            run = used[dup_off[p]:dup_off[p + 1]]
            k = len(run) if ci % 2 == 1 else (0 if not run.any() else int(np.max(np.nonzero(run)[0])) + 1)
            keep[dup_off[p]:dup_off[p] + k] = True
            mult_c[p] = k
        dup_c = np.concatenate([[0], np.cumsum(mult_c)]).astype(np.uint32)
        E_c = int(dup_c[-1])
        stride = P + 5 + ci  # padded
        prim = np.zeros((n, stride, 8), np.int32)
        prim[:, :P] = chunk[:, :P]
        ext_stride = E_c + 3
        ext = np.zeros((n, max(ext_stride, 1), 8), np.int32)
        ext[:, :, 0] = ABSENT
        if E_c:
            ext[:, :E_c] = chunk[:, P:][:, keep]
        d_prim, d_ext = _pack(ctx, prim, layout), _pack(ctx, ext, layout)
        rec = ctx.records(d_prim, layout, n, E=E_c, row_stride=stride, ext=d_ext, ext_stride=ext_stride, dup_off=_t(dup_c))
        last = ci == len(cuts) - 2
        fin = ctx.error_reduce_records(rec, P, acc, 0.002, 100, first_sample=lo, accumulate=ci > 0, finalize=last)
    ctx.set_tuning(0)
    assert_acc_equal(acc, ref, skip=("gm_first",) if splits else ())
    assert_final_equal(fin, ref_fin)


@pytest.mark.parametrize("layout", ["i32", "u24"])
def test_dense_rows_inside_one_array_with_a_padded_stride(ctx, layout):
    """ext == NULL: the extras follow the primaries inside each (padded) row."""
    rng = np.random.default_rng(5)
    P, S = 500, 9
    recs, E, dup_off = _panel_with_extras(P, S, rng)
    stride = P + E + 64
    padded = np.zeros((S, stride, 8), np.int32)
    padded[:, :P + E] = recs
    ref = orc.error_reduce(recs, P, 0.002, 100, E=E, dup_off=dup_off)
    acc = ctx.new_acc(P)
    rec = ctx.records(_pack(ctx, padded, layout), layout, S, E=E, row_stride=stride, dup_off=_t(dup_off))
    ctx.error_reduce_records(rec, P, acc, 0.002, 100)
    assert_acc_equal(acc, ref)


@pytest.mark.parametrize("layout", ["i32", "u24", "u16"])
def test_poisson_call_over_a_described_cohort(ctx, layout):
    from amplisolve_amd.api import POISSON_FULL, POISSON_PREFILTER

    rng = np.random.default_rng(17)
    P, T, E = 700, 11, 53
    trecs = edge_case_recs(P + E, T, rng)
    trecs[:, ::7, :] = np.array([30, 0, 0, 400, 25, 0, 0, 380], np.int32)
    if layout == "u16":
        trecs = np.where(trecs == ABSENT, ABSENT, np.minimum(trecs, 65534)).astype(np.int32)
    thr = rng.choice(np.array([0.002, 0.01, 0.0, -1.0, 0.000731, 0.05, -2.0], np.float32), size=(2, 4, P)).astype(np.float32)
    ref_code = rng.integers(0, 4, P).astype(np.uint8)
    ref_code[::11] = 255
    ext_pos = rng.integers(0, P, E).astype(np.uint32)
    exp = orc.poisson_call(trecs, P, thr, ref_code, 100, E=E, ext_pos=ext_pos)
    assert exp["call_mask"].any()
    stride, ext_stride = P + 9, E + 2
    prim = np.zeros((T, stride, 8), np.int32)
    prim[:, :P] = trecs[:, :P]
    ext = np.zeros((T, ext_stride, 8), np.int32)
    ext[:, :E] = trecs[:, P:]
    rec = ctx.records(_pack(ctx, prim, layout), layout, T, E=E, row_stride=stride, ext=_pack(ctx, ext, layout), ext_stride=ext_stride,
                      ext_pos=_t(ext_pos))
    for mode in (POISSON_FULL, POISSON_PREFILTER):
        res = ctx.poisson_call_records(rec, P, _t(thr), _t(ref_code), 100, mode=mode, capacity=4 * (P + E) * T + 64)
        assert np.array_equal(res["call_mask"].cpu().numpy(), exp["call_mask"])
        calls = ctx.read_calls(res)
        assert len(calls) == sum(bin(int(v)).count("1") for v in exp["call_mask"].ravel())
        for c in calls:  # the evidence the annotation needs travels with the call
            r = trecs[c["sample"], c["record"]]
            assert (c["k_fw"], c["k_bw"]) == (r[c["alt"]], r[4 + c["alt"]])
            assert (c["fw"], c["bw"], c["rd"]) == (r[:4].sum(), r[4:].sum(), r.sum())


@pytest.mark.parametrize("rows,blocks", [(1, 1), (2, 3), (3, 32), (6, 0), (24, 7), (1000, 1)])
def test_poisson_launch_shape_never_changes_a_result(ctx, rows, blocks):
    """rows per wave / drain workgroups per shard: same mask, same calls, Q within 1e-6 of the oracle."""
    from amplisolve_amd.api import POISSON_PREFILTER

    P, T = 3000, 29
    normals = synth_recs(P, 24)
    fin = orc.error_finalize(orc.error_reduce(normals, P))
    ref_code = synth_ref(P)
    trecs = synth_recs(P, T, tumour=True)
    exp = orc.poisson_call(trecs, P, fin["thr"], ref_code, 100)
    ctx.set_poisson_tuning(rows, blocks)
    try:
        res = ctx.poisson_call(_t(trecs), P, _t(fin["thr"]), _t(ref_code), 100, mode=POISSON_PREFILTER, capacity=4 * P * T + 64)
    finally:
        ctx.set_poisson_tuning()
    assert np.array_equal(res["call_mask"].cpu().numpy(), exp["call_mask"])
    calls = ctx.read_calls(res)
    assert len(calls) == sum(bin(int(v)).count("1") for v in exp["call_mask"].ravel()) and len(calls) > 0
    qo = exp["q"]
    for c in calls:
        for st, name in enumerate(("q_fw", "q_bw")):
            e = qo[c["sample"], c["record"], c["alt"], st]
            assert abs(c[name] - e) <= 1e-6 * max(1.0, abs(e))


@pytest.mark.parametrize("layout", ["i32", "u24"])
def test_lines_with_their_own_rd_column(ctx, layout):
    """RD != A+C+G+T (EE:1178-1181, VC:762-765): the column travels as a plane beside the records and decides where the
    reference lets it -- Germ_Max AF = X / RD (EE:1229-1232), AF = X / RD and forward depth RD - RD_reverse in the caller
    (VC:814-817, VC:895) -- incl. RD = 0, RD below the reverse depth, RD far above the sum."""
    from amplisolve_amd.api import POISSON_FULL, POISSON_PREFILTER

    rng = np.random.default_rng(23)
    P, S = 600, 21
    recs, E, dup_off = _panel_with_extras(P, S, rng)
    R = P + E
    rd = np.full((S, R), ABSENT, np.int32)
    pick = rng.random((S, R)) < 0.2
    tot = np.where(recs[:, :, 0] == ABSENT, 0, recs.sum(-1))
    choice = rng.integers(0, 7, (S, R))
    alt = np.select([choice == 0, choice == 1, choice == 2, choice == 3, choice == 4, choice == 5], [tot * 2, tot + 37, tot // 2, tot // 3, 0 * tot, 7 + 0 * tot], tot * 40)
    rd[pick] = alt[pick].astype(np.int32)
    ref = orc.error_reduce(recs, P, 0.002, 100, E=E, dup_off=dup_off, rd=rd)
    assert not np.array_equal(ref["gm_n"], orc.error_reduce(recs, P, 0.002, 100, E=E, dup_off=dup_off)["gm_n"])
    prim = np.ascontiguousarray(recs[:, :P])
    ext = np.ascontiguousarray(recs[:, P:])
    rec = ctx.records(_pack(ctx, prim, layout), layout, S, E=E, ext=_pack(ctx, ext, layout), dup_off=_t(dup_off),
                      rd=_t(np.ascontiguousarray(rd[:, :P])), rd_ext=_t(np.ascontiguousarray(rd[:, P:])))
    acc = ctx.new_acc(P)
    fin = ctx.error_reduce_records(rec, P, acc, 0.002, 100, finalize=True)
    assert ctx.flags() == 0
    assert_acc_equal(acc, ref)
    assert_final_equal(fin, orc.error_finalize(ref))
    # the calling half on the same kind of cohort
    T = 9
    trecs = edge_case_recs(R, T, rng)
    trecs[:, ::5, :] = np.array([30, 0, 0, 400, 25, 0, 0, 380], np.int32)
    trd = np.full((T, R), ABSENT, np.int32)
    tp = rng.random((T, R)) < 0.3
    ttot = np.where(trecs[:, :, 0] == ABSENT, 0, trecs.sum(-1))
    tch = rng.integers(0, 6, (T, R))
    talt = np.select([tch == 0, tch == 1, tch == 2, tch == 3, tch == 4], [ttot * 2, ttot + 11, ttot // 2, 0 * ttot, trecs[:, :, 4:].sum(-1) - 3], ttot * 9)
    trd[tp] = talt[tp].astype(np.int32)
    thr = rng.choice(np.array([0.002, 0.01, 0.0, -1.0, 0.000731, 0.05], np.float32), size=(2, 4, P)).astype(np.float32)
    ref_code = rng.integers(0, 4, P).astype(np.uint8)
    ext_pos = np.repeat(np.arange(P), np.diff(dup_off.astype(np.int64))).astype(np.uint32)
    exp = orc.poisson_call(trecs, P, thr, ref_code, 100, E=E, ext_pos=ext_pos, rd=trd)
    exp0 = orc.poisson_call(trecs, P, thr, ref_code, 100, E=E, ext_pos=ext_pos)
    assert exp["call_mask"].any() and not np.array_equal(exp["call_mask"], exp0["call_mask"])
    trec = ctx.records(_pack(ctx, np.ascontiguousarray(trecs[:, :P]), layout), layout, T, E=E, ext=_pack(ctx, np.ascontiguousarray(trecs[:, P:]), layout),
                       ext_pos=_t(ext_pos), rd=_t(np.ascontiguousarray(trd[:, :P])), rd_ext=_t(np.ascontiguousarray(trd[:, P:])))
    for mode in (POISSON_FULL, POISSON_PREFILTER):
        res = ctx.poisson_call_records(trec, P, _t(thr), _t(ref_code), 100, mode=mode, capacity=32 * 3 * R * T)  # every call could land in one segment
        assert np.array_equal(res["call_mask"].cpu().numpy(), exp["call_mask"])
        for c in ctx.read_calls(res):
            r = trecs[c["sample"], c["record"]]
            want_rd = trd[c["sample"], c["record"]] if trd[c["sample"], c["record"]] != ABSENT else r.sum()
            assert c["rd"] == want_rd and np.float32(c["af"]) == exp["af"][c["sample"], c["record"], c["alt"], 0]


def test_calls_within_rounding_of_the_gate_are_flagged_and_decided_by_the_guard(ctx):
    """(k, RD, err) whose REFERENCE Q lies within 1e-6 of the gate Q >= 5, on both sides of it: the device cannot decide
    them (its exp / log differ from glibc's in the last bits), so it lists every one of them with AMPLI_CALL_BORDERLINE --
    above or below -- and the host's guard, which repeats the reference's operation sequence, decides as the reference."""
    import ctypes as C

    from amplisolve_amd import host_lib
    from amplisolve_amd.api import POISSON_FULL, POISSON_PREFILTER
    from tests.helpers import borderline_triples

    tri = borderline_triples(16)
    assert len(tri) >= 8 and any(q >= 5 for *_, q in tri) and any(q < 5 for *_, q in tri)
    P, T = len(tri), 2
    trecs = np.zeros((T, P, 8), np.int32)
    thr = np.full((2, 4, P), 0.002, np.float32)
    for i, (k, rd, err, _) in enumerate(tri):
        trecs[:, i] = [rd - k, k, 0, 0, 950, 50, 0, 0]  # reference A; alt C: forward strand on the knife edge, reverse Q = 100
        thr[0, 1, i] = err
    ref_code = np.zeros(P, np.uint8)
    exp = orc.poisson_call(trecs, P, thr, ref_code, 100)
    want = {(t, i): bool(exp["call_mask"][t, i] >> 1 & 1) for t in range(T) for i in range(P)}
    assert any(want.values()) and not all(want.values())
    H = host_lib()
    for mode in (POISSON_PREFILTER, POISSON_FULL):
        res = ctx.poisson_call(_t(trecs), P, _t(thr), _t(ref_code), 100, mode=mode, capacity=32 * 4 * P * T)
        calls = [c for c in ctx.read_calls(res) if c["alt"] == 1]
        assert {(c["sample"], c["record"]) for c in calls} == set(want)  # every pair is listed, called or not
        for c in calls:
            assert c["flags"] & 1  # AMPLI_CALL_BORDERLINE
            k, rd, err, q = tri[c["record"]]
            assert abs(c["q_fw"] - q) < 1e-6 and c["q_bw"] == 100.0 and (c["k_fw"], c["rd"] - c["bw"]) == (k, rd)
            ge5 = C.c_int32()
            got = H.ampli_host_guard_score(int(c["k_fw"]), int(c["rd"] - c["bw"]), float(thr[0, 1, c["record"]]), C.byref(ge5), None)
            assert got == q and bool(ge5.value) == want[(c["sample"], c["record"])]
