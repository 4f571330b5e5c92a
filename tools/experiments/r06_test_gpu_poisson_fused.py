"""poisson_call's prefilter mode as ONE kernel (poisson_fused_kernel, the default since round 6) against the two-kernel form of rounds
1-5 (stream kernel + queue in HBM + drain kernel) and against the oracle: call mask byte for byte, call list entry for entry (Q bit for
bit: the same scorer code on the same inputs), call counts per use -- over every shape that takes another path through the kernel:
positions listed twice (extras), lines with their own RD column, rows not a multiple of the wave's run, panels whose length is not a
multiple of four (byte stores of the mask), all three record layouts, position ranges on two streams, panels where nearly every record
survives the no-call bound (a wave's 64 staging slots fill up mid-row), repeated calls on one context (the counters the last workgroup
leaves clean), counting without a list, no list at all."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tests.helpers import edge_case_recs, synth_recs, synth_ref
from tests.test_gpu_parity import _t
from tests.test_gpu_records import _pack

pytestmark = pytest.mark.gpu
ABSENT = np.iinfo(np.int32).min
FIELDS = ("sample", "record", "alt", "rd", "q_fw", "q_bw", "af", "af_fw", "af_bw", "k_fw", "k_bw", "fw", "bw", "flags")


def _both(ctx, call):
    """call(ctx) under the one-kernel and the two-kernel form"""
    from amplisolve_amd.api import POISSON_PREFILTER  # noqa: F401

    one = call()
    ctx.set_poisson_fused(False)
    try:
        two = call()
    finally:
        ctx.set_poisson_fused(True)
    return one, two


def _assert_same(ctx, one, two, exp_mask=None):
    import torch

    assert torch.equal(one["call_mask"], two["call_mask"])
    if exp_mask is not None:
        assert np.array_equal(one["call_mask"].cpu().numpy(), exp_mask)
    if one["n_calls"] is not None:
        assert ctx.n_calls_total(one) == ctx.n_calls_total(two)
    if one["calls_buf"] is not None:
        a, b = ctx.read_calls(one), ctx.read_calls(two)
        assert len(a) == len(b)
        for f in FIELDS:
            assert np.array_equal(a[f], b[f]), f
        return a
    return None


@pytest.mark.parametrize("layout", ["i32", "u24", "u16"])
@pytest.mark.parametrize("P,T,E,rows", [(700, 11, 53, 0), (701, 29, 0, 0), (64, 1, 0, 0), (1000, 97, 7, 0), (3001, 5, 130, 3), (130, 40, 0, 32), (4096, 13, 0, 1)])
def test_one_kernel_equals_two_kernels_and_the_oracle(ctx, layout, P, T, E, rows):
    from amplisolve_amd.api import POISSON_PREFILTER

    rng = np.random.default_rng(P * 31 + T)
    trecs = edge_case_recs(P + E, T, rng)
    trecs[:, ::7, :] = np.array([30, 0, 0, 400, 25, 0, 0, 380], np.int32)
    if layout == "u16":
        trecs = np.where(trecs == ABSENT, ABSENT, np.minimum(trecs, 65534)).astype(np.int32)
    thr = rng.choice(np.array([0.002, 0.01, 0.0, -1.0, 0.000731, 0.05, -2.0], np.float32), size=(2, 4, P)).astype(np.float32)
    ref_code = rng.integers(0, 4, P).astype(np.uint8)
    ref_code[::11] = 255
    ext_pos = rng.integers(0, P, E).astype(np.uint32) if E else None
    exp = orc.poisson_call(trecs, P, thr, ref_code, 100, E=E, ext_pos=ext_pos)
    stride, ext_stride = P + 9, E + 2
    prim = np.zeros((T, stride, 8), np.int32)
    prim[:, :P] = trecs[:, :P]
    ext = np.zeros((T, ext_stride, 8), np.int32)
    ext[:, :E] = trecs[:, P:]
    rec = ctx.records(_pack(ctx, prim, layout), layout, T, E=E, row_stride=stride, ext=_pack(ctx, ext, layout) if E else None, ext_stride=ext_stride if E else 0,
                      ext_pos=_t(ext_pos) if E else None)
    ctx.set_poisson_tuning(rows, 0)
    try:
        one, two = _both(ctx, lambda: ctx.poisson_call_records(rec, P, _t(thr), _t(ref_code), 100, mode=POISSON_PREFILTER, capacity=32 * 3 * (P + E) * T + 64))
    finally:
        ctx.set_poisson_tuning()
    calls = _assert_same(ctx, one, two, exp["call_mask"])
    assert ctx.flags() == 0
    assert len(calls) >= sum(bin(int(v)).count("1") for v in exp["call_mask"].ravel())  # + pairs within 1e-6 of the gate, flagged
    qo = exp["q"]
    for c in calls[:: max(1, len(calls) // 300)]:
        for st, name in enumerate(("q_fw", "q_bw")):
            e = qo[c["sample"], c["record"], c["alt"], st]
            assert abs(c[name] - e) <= 1e-6 * max(1.0, abs(e))


def test_lines_with_their_own_rd_column(ctx):
    from amplisolve_amd.api import POISSON_PREFILTER

    rng = np.random.default_rng(5)
    P, T, E = 900, 17, 40
    R = P + E
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
    ext_pos = rng.integers(0, P, E).astype(np.uint32)
    exp = orc.poisson_call(trecs, P, thr, ref_code, 100, E=E, ext_pos=ext_pos, rd=trd)
    trec = ctx.records(_t(np.ascontiguousarray(trecs[:, :P])), "i32", T, E=E, ext=_t(np.ascontiguousarray(trecs[:, P:])), ext_pos=_t(ext_pos),
                       rd=_t(np.ascontiguousarray(trd[:, :P])), rd_ext=_t(np.ascontiguousarray(trd[:, P:])))
    one, two = _both(ctx, lambda: ctx.poisson_call_records(trec, P, _t(thr), _t(ref_code), 100, mode=POISSON_PREFILTER, capacity=32 * 3 * R * T))
    calls = _assert_same(ctx, one, two, exp["call_mask"])
    for c in calls:
        r = trecs[c["sample"], c["record"]]
        want_rd = trd[c["sample"], c["record"]] if trd[c["sample"], c["record"]] != ABSENT else r.sum()
        assert c["rd"] == want_rd


@pytest.mark.parametrize("P,T", [(70_000, 4), (999, 50)])
def test_panels_where_every_record_survives_the_bound(ctx, P, T):
    """strong variants on both strands of every record: each wave's 64 staging slots fill up inside its first row and the wave
    scores them in place, over and over; 3 calls per record"""
    from amplisolve_amd.api import POISSON_PREFILTER

    rng = np.random.default_rng(17)
    trecs = np.zeros((T, P, 8), np.int32)
    trecs[:, :, 0] = 900; trecs[:, :, 4] = 850
    for a in (1, 2, 3):
        trecs[:, :, a] = rng.integers(10, 60, (T, P)); trecs[:, :, 4 + a] = rng.integers(10, 60, (T, P))
    thr = np.full((2, 4, P), 0.002, np.float32)
    ref_code = np.zeros(P, np.uint8)
    ctx.set_queue_items(3 * T * P)
    try:
        one, two = _both(ctx, lambda: ctx.poisson_call(_t(trecs), P, _t(thr), _t(ref_code), 100, mode=POISSON_PREFILTER, capacity=4 * 3 * T * P))
    finally:
        ctx.set_queue_items(0)
    calls = _assert_same(ctx, one, two)
    assert ctx.flags() == 0 and len(calls) == 3 * T * P and ctx.n_calls_total(one) == 3 * T * P
    assert (one["call_mask"].cpu().numpy() == 0b1110).all()


def test_repeated_calls_counts_only_and_no_list(ctx):
    """the context's counters are left clean by the last workgroup of every launch: call after call gives the same counts, with a
    list, counting only (n_calls without a list) and with neither"""
    import torch

    from amplisolve_amd.api import CALL_COUNTER_WORDS, POISSON_PREFILTER

    P, S, T = 20_000, 24, 21
    fin = ctx.error_estimate(ctx.synth_fill(P, S), P)
    tum, refc = ctx.synth_fill(P, T, tumour=True), ctx.synth_ref(P)
    first = ctx.poisson_call(tum, P, fin.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 18)
    n = ctx.n_calls_total(first)
    assert n > 0
    for _ in range(4):
        again = ctx.poisson_call(tum, P, fin.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 18)
        assert ctx.n_calls_total(again) == n and torch.equal(again["call_mask"], first["call_mask"])
    counting = ctx.poisson_call(tum, P, fin.thr, refc, 100, mode=POISSON_PREFILTER, n_calls=torch.full((CALL_COUNTER_WORDS,), 77, dtype=torch.int64, device="cuda"))
    assert ctx.n_calls_total(counting) == n and counting["calls_buf"] is None
    bare = ctx.poisson_call(tum, P, fin.thr, refc, 100, mode=POISSON_PREFILTER)
    assert bare["n_calls"] is None and torch.equal(bare["call_mask"], first["call_mask"])
    after = ctx.poisson_call(tum[:5], P, fin.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 18)  # another grid size: another last workgroup
    sub = ctx.read_calls(after)
    full = ctx.read_calls(first)
    assert len(sub) == int((full["sample"] < 5).sum()) > 0


@pytest.mark.parametrize("P", [100_000, 99_968])
def test_position_ranges_on_two_streams(ctx, P):
    """ampli_set_ranges(2): each range's kernel owns its half of the call list's shards and of every mask row"""
    import torch

    from amplisolve_amd.api import POISSON_PREFILTER

    S, T = 16, 12
    normals, tum, refc = ctx.synth_fill(P, S), ctx.synth_fill(P, T, tumour=True), ctx.synth_ref(P)
    fin = ctx.error_estimate(normals, P)
    plain = ctx.poisson_call(tum, P, fin.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 18)
    ctx.set_ranges(2)
    try:
        for _ in range(3):
            fin2 = ctx.error_estimate(normals, P)
            res = ctx.poisson_call(tum, P, fin2.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 18)
        ctx.ranges_join()
    finally:
        ctx.set_ranges(1)
    assert torch.equal(res["call_mask"], plain["call_mask"])
    a, b = ctx.read_calls(res), ctx.read_calls(plain)
    assert len(a) == len(b) > 0
    for f in FIELDS:
        assert np.array_equal(a[f], b[f]), f
    assert ctx.flags() == 0
