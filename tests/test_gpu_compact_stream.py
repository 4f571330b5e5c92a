"""error_reduce_u16_kernel beyond the headline's resident, listed-once shape (round 5): cohorts that list positions more than
once (E > 0: every line of a key is a record, EE:1555; the Germ_Max order of EE:1251-1271 is primary, extras, next sample),
cohorts streamed chunk by chunk through an accumulator table taken as streaming state (AMPLI_REDUCE_SUMMARY), and the last
chunk of a shard written slice-major -- the shapes the two command lines launch.  Oracle = oracle/ampli_oracle.c."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tests.helpers import edge_case_recs, synth_recs
from tests.test_gpu_parity import _t, assert_acc_equal, assert_final_equal
from tests.test_gpu_records import _pack

pytestmark = pytest.mark.gpu
ABSENT = np.iinfo(np.int32).min
COMPACT, GENERAL = {"u16": "error_reduce_u16_kernel", "u24": "error_reduce_u24_kernel"}, "error_reduce_kernel"


def _cohort(P, S, rng, dup_frac=8, triple_frac=50, runs=True):
    """uint16-range records with edge cases mixed in; positions listed twice / three times, as runs (overlapping amplicons list
    a stretch of consecutive positions twice) and scattered ones, incl. the first and the last position and tile boundaries"""
    mult = np.zeros(P, np.int64)
    if P >= 8:
        mult[rng.choice(P, max(1, P // dup_frac), replace=False)] = 1
        mult[rng.choice(P, max(1, P // triple_frac), replace=False)] = 2
        if runs and P > 200:
            a = int(rng.integers(0, P - 150))
            mult[a:a + 140] = 1
    mult[0] = 1
    mult[P - 1] = 2
    if P > 64:
        mult[63] = mult[64] = 1
    dup_off = np.concatenate([[0], np.cumsum(mult)]).astype(np.uint32)
    E = int(dup_off[-1])
    recs = np.concatenate([synth_recs(P, S), edge_case_recs(E, S, rng)], axis=1)
    if P >= 63:
        e = edge_case_recs(P, S, rng)
        pick = rng.random((S, P)) < 0.25
        recs[:, :P][pick] = e[pick]
    recs = np.where(recs == ABSENT, ABSENT, np.minimum(recs, 65534)).astype(np.int32)
    return recs, E, dup_off


def _summary_equal(acc, ref):
    """a table written as streaming state: every additive plane exact; gm_n as none / one / more; AFs where they are defined"""
    pl = {k: v.cpu().numpy() for k, v in acc.planes().items()}
    for name in ("snt", "srd", "cnt", "nrec"):
        a, b = pl[name], ref[name]
        assert np.array_equal(a.view(np.int64 if a.dtype.itemsize == 8 else np.int32), b.view(np.int64 if b.dtype.itemsize == 8 else np.int32)), name
    assert np.array_equal(pl["gm_n"] > 0, ref["gm_n"] > 0) and np.array_equal(pl["gm_n"] > 1, ref["gm_n"] > 1)
    m1, m2 = ref["gm_n"] > 0, ref["gm_n"] > 1
    assert np.array_equal(pl["gm_first_af"][m1].view(np.int32), ref["gm_first_af"][m1].view(np.int32))
    assert np.array_equal(pl["gm_rest"][m2].view(np.int32), ref["gm_rest"][m2].view(np.int32))
    assert np.all((pl["gm_first"][m1] == ref["gm_first"][m1]) | (pl["gm_first"][m1] == -1))
    assert np.all(pl["gm_first"][~m1] == np.iinfo(np.int32).max)


@pytest.mark.parametrize("lay", ["u16", "u24"])
@pytest.mark.parametrize("P,S", [(1, 1), (64, 3), (65, 4), (130, 9), (1000, 33), (4097, 37), (777, 130), (20000, 64), (300, 700)])
def test_positions_listed_more_than_once_through_the_compact_kernel(ctx, P, S, lay):
    """E > 0: the compact kernel keeps every tile of positions listed once; the tiles that hold a position listed twice or more go,
    whole, to error_reduce_kernel over dup_tiles_kernel's list.  Same table as the oracle and as the general kernel alone."""
    import torch

    rng = np.random.default_rng(P * 7 + S)
    recs, E, dup_off = _cohort(P, S, rng)
    want = orc.error_finalize(orc.error_reduce(recs, P, 0.002, 100, E=E, dup_off=dup_off))
    ctx.set_record_layout(lay)
    try:
        packed = _pack(ctx, recs, lay)
        outs = []
        for compact in (True, False):
            ctx.set_reduce_compact(compact)
            ctx.set_tuning(1 if compact else 0, groups=1 if compact else 0)
            got = ctx.error_estimate(packed, P, 0.002, 100, E=E, dup_off=_t(dup_off))
            assert ctx.flags() == 0
            assert ctx.last_reduce_kernel() == (COMPACT[lay] if compact else GENERAL)
            assert_final_equal(got, want)
            outs.append(got)
        for k in ("rate", "thr", "code", "germ_present"):
            assert torch.equal(getattr(outs[0], k).view(torch.uint8), getattr(outs[1], k).view(torch.uint8)), k
    finally:
        ctx.set_reduce_compact(True)
        ctx.set_tuning(0)
        ctx.set_record_layout("i32")


@pytest.mark.parametrize("P,S,cuts,extras", [(300, 13, (0, 13), True), (300, 13, (0, 1, 2, 13), True), (1000, 40, (0, 7, 8, 29, 40), False),
                                             (4097, 37, (0, 16, 32, 37), True), (20000, 96, (0, 32, 64, 96), False), (65, 600, (0, 300, 600), True)])
@pytest.mark.parametrize("mix", ["compact", "general_first", "general_last", "u24_compact"])
def test_streamed_chunks_through_the_compact_kernel(ctx, P, S, cuts, extras, mix):
    """What AmpliSolveErrorEstimation launches: a uint16 cohort in chunks of consecutive samples, each chunk its own buffers, folded
    into ONE table taken as streaming state; the last launch finalises.  Every launch is the compact kernel (asserted); with `mix`
    one chunk takes the general kernel instead (as a chunk in another layout would), before or after compact ones: the exact planes
    and the summary compose in either order.  Table of the single pass, error table of the oracle."""
    rng = np.random.default_rng(P + S + len(cuts))
    if extras:
        recs, E, dup_off = _cohort(P, S, rng)
    else:
        recs, E, dup_off = _cohort(P, S, rng)[0][:, :P], 0, np.zeros(P + 1, np.uint32)
    ref = orc.error_reduce(recs, P, 0.002, 100, E=E, dup_off=dup_off)
    ref_fin = orc.error_finalize(ref)
    acc = ctx.new_acc(P)
    acc.buf.fill_(0x5A)
    ctx.set_tuning(1, groups=1)
    fin = None
    n_chunks = len(cuts) - 1
    try:
        for ci in range(n_chunks):
            lo, hi = cuts[ci], cuts[ci + 1]
            chunk = recs[lo:hi]
            stride = P + 3 + ci
            prim = np.zeros((hi - lo, stride, 8), np.int32)
            prim[:, :P] = chunk[:, :P]
            ext = np.zeros((hi - lo, max(E, 1) + 2, 8), np.int32)
            ext[:, :, 0] = ABSENT
            if E:
                ext[:, :E] = chunk[:, P:]
            # "u24_compact": every other chunk arrives in 24-bit records (a chunk whose counts need them): the two compact kernels share the table
            lay = "u24" if (mix == "u24_compact" and ci % 2 == 1) else "u16"
            rec = ctx.records(_pack(ctx, prim, lay), lay, hi - lo, E=E, row_stride=stride, ext=_pack(ctx, ext, lay), ext_stride=max(E, 1) + 2,
                              dup_off=_t(dup_off))
            general = n_chunks > 1 and ((mix == "general_first" and ci == 0) or (mix == "general_last" and ci == n_chunks - 1))
            ctx.set_reduce_compact(not general)
            fin = ctx.error_reduce_records(rec, P, acc, 0.002, 100, first_sample=lo, accumulate=ci > 0, finalize=ci == n_chunks - 1, summary=True)
            assert ctx.last_reduce_kernel() == (GENERAL if general else COMPACT[lay])
        assert ctx.flags() == 0
    finally:
        ctx.set_reduce_compact(True)
        ctx.set_tuning(0)
    _summary_equal(acc, ref)
    assert_final_equal(fin, ref_fin)


def test_a_table_of_record_still_gets_exact_planes(ctx):
    """without AMPLI_REDUCE_SUMMARY a table is the bookkeeping of record: the general kernel writes it (gm_n exact, gm_first known)"""
    P, S = 1000, 24
    recs = np.minimum(synth_recs(P, S), 65534)
    ref = orc.error_reduce(recs, P, 0.002, 100)
    acc = ctx.new_acc(P)
    ctx.set_tuning(1, groups=1)
    try:
        rec = ctx.records(_pack(ctx, recs, "u16"), "u16", S)
        ctx.error_reduce_records(rec, P, acc, 0.002, 100)
        assert ctx.last_reduce_kernel() == GENERAL
    finally:
        ctx.set_tuning(0)
    assert_acc_equal(acc, ref)


@pytest.mark.parametrize("slim", [False, True])
@pytest.mark.parametrize("P,S,n,cuts,extras", [(4097, 64, 4, (0, 20, 40, 64), True), (1000, 37, 3, (0, 37), False), (20000, 96, 2, (0, 48, 96), False)])
def test_last_chunk_of_a_shard_goes_slice_major(ctx, P, S, n, cuts, extras, slim):
    """The sharded command line: a shard's chunks fold into its table and the LAST launch writes table (+) chunk straight into the
    slice-major exchange buffers (ampli_error_reduce_records_sliced; no table at all for a one-chunk shard).  Buffers equal to
    ampli_error_reduce_sliced over the shard's resident records through the general kernel."""
    import torch

    from amplisolve_amd.dist import slice_geometry, slice_planes

    rng = np.random.default_rng(P + S + n)
    if extras:
        recs, E, dup_off = _cohort(P, S, rng)
    else:
        recs, E, dup_off = _cohort(P, S, rng)[0][:, :P], 0, np.zeros(P + 1, np.uint32)
    L, _, _, _ = slice_geometry(P, n, slim)
    pl = slice_planes(slim)
    ctx.set_record_layout("u16")
    ctx.set_slice_format(slim)
    try:
        ctx.set_reduce_compact(False)
        want_s = torch.zeros(n * pl * L, dtype=torch.float64, device="cuda")
        want_g = torch.zeros(n * 8 * L, dtype=torch.float32, device="cuda")
        ctx.error_reduce_sliced(_pack(ctx, recs, "u16"), P, n, want_s, want_g, E=E, dup_off=_t(dup_off), first_sample=5)
        ctx.set_reduce_compact(True)
        ctx.set_tuning(1, groups=1)
        got_s, got_g = torch.zeros_like(want_s), torch.zeros_like(want_g)
        acc = ctx.new_acc(P) if len(cuts) > 2 else None
        for ci in range(len(cuts) - 1):
            lo, hi = cuts[ci], cuts[ci + 1]
            chunk = recs[lo:hi]
            ext = np.zeros((hi - lo, max(E, 1), 8), np.int32)
            ext[:, :, 0] = ABSENT
            if E:
                ext[:, :E] = chunk[:, P:]
            rec = ctx.records(_pack(ctx, np.ascontiguousarray(chunk[:, :P]), "u16"), "u16", hi - lo, E=E, ext=_pack(ctx, ext, "u16"), dup_off=_t(dup_off))
            if ci == len(cuts) - 2:
                ctx.error_reduce_records_sliced(rec, P, acc, n, got_s, got_g, first_sample=5 + lo, accumulate=ci > 0)
            else:
                ctx.error_reduce_records(rec, P, acc, 0.002, 100, first_sample=5 + lo, accumulate=ci > 0, summary=True)
            assert ctx.last_reduce_kernel() == COMPACT["u16"]
        assert ctx.flags() == 0
        assert torch.equal(got_s.view(torch.int64), want_s.view(torch.int64))
        assert torch.equal(got_g.view(torch.int32), want_g.view(torch.int32))
    finally:
        ctx.set_reduce_compact(True)
        ctx.set_tuning(0)
        ctx.set_slice_format(False)
        ctx.set_record_layout("i32")


def test_streamed_chunks_over_position_ranges(ctx):
    """The table carry (TAB) and the position ranges together: a uint16 cohort in three chunks through the summary table with the
    launches cut into two / three position ranges (each range reads and writes its own positions of the table); the last launch
    finalises.  Error table of the oracle; a joining call (flags) before the table is looked at."""
    P, S, cuts = 20_000, 48, (0, 16, 32, 48)
    rng = np.random.default_rng(7)
    recs = _cohort(P, S, rng)[0][:, :P]
    ref = orc.error_reduce(recs, P, 0.002, 100)
    for n in (2, 3):
        acc = ctx.new_acc(P)
        ctx.set_tuning(1, groups=1)
        ctx.set_ranges(n)
        try:
            fin = None
            for ci in range(3):
                lo, hi = cuts[ci], cuts[ci + 1]
                rec = ctx.records(_pack(ctx, np.ascontiguousarray(recs[lo:hi]), "u16"), "u16", hi - lo)
                fin = ctx.error_reduce_records(rec, P, acc, 0.002, 100, first_sample=lo, accumulate=ci > 0, finalize=ci == 2, summary=True)
                assert ctx.last_reduce_kernel() == COMPACT["u16"]
            assert ctx.flags() == 0  # joins the ranges
        finally:
            ctx.set_ranges(1)
            ctx.set_tuning(0)
        _summary_equal(acc, ref)
        assert_final_equal(fin, orc.error_finalize(ref))


def test_u24_strand_depth_sums_beyond_2_to_31_per_lane(ctx):
    """a lane of error_reduce_u24_kernel takes up to 1023 records whose strand depths lie just under 2^22, so its 32-bit depth
    sums pass INT_MAX and reach towards 2^32: kept unsigned and widened at the end.  Same table as the oracle, no rerun flag."""
    rng = np.random.default_rng(2)
    P, S = 130, 2400  # four waves of 600 records: 600 x 4.19e6 = 2.5e9 > 2^31 per lane and strand
    recs = np.zeros((S, P, 8), np.int32)
    fwd = rng.random(P) < 0.5  # which strand carries the depth at this position (RD = FW + BW stays below 2^22)
    big = rng.integers(4_190_000, 4_194_000, (S, P))
    small = rng.integers(100, 300, (S, P))
    FW = np.where(fwd[None, :], big, small)
    BW = np.where(fwd[None, :], small, big)
    for st, depth in ((0, FW), (4, BW)):
        alts = (depth[:, :, None] * rng.uniform(0, 0.012, (S, P, 3))).astype(np.int64)
        recs[:, :, st + 1: st + 4] = alts
        recs[:, :, st] = depth - alts.sum(-1)
    assert int((recs[:, :, :4].sum(-1) + recs[:, :, 4:].sum(-1)).max()) < (1 << 22)
    ref = orc.error_reduce(recs, P, 0.002, 100)
    assert int(ref["srd"].max()) > (1 << 31) * 4 and ref["order_sensitive"] == 0
    want = orc.error_finalize(ref)
    ctx.set_record_layout("u24")
    try:
        packed = _pack(ctx, recs, "u24")
        ctx.set_reduce_compact(True)
        ctx.set_tuning(1, groups=1)
        got = ctx.error_estimate(packed, P, 0.002, 100)
        assert ctx.flags() == 0 and ctx.last_reduce_kernel() == COMPACT["u24"]
        assert_final_equal(got, want)
        acc = ctx.new_acc(P)
        ctx.set_reduce_compact(False)
        ctx.set_tuning(0, groups=0)
        fin = ctx.error_estimate(packed, P, 0.002, 100, acc=acc)
        assert_acc_equal(acc, ref)
        assert_final_equal(fin, want)
    finally:
        ctx.set_reduce_compact(True)
        ctx.set_tuning()
        ctx.set_record_layout("i32")


@pytest.mark.parametrize("where", ["first", "middle", "last"])
def test_u24_depth_between_2_to_22_and_2_to_24_raises_the_rerun_flag(ctx, where):
    """a covered 24-bit record with RD in [2^22, 2^24) is packable (so the compact 24-bit kernel is chosen) but outside the
    fast arithmetic: the kernel's only guard is its DEPTH_CHECK bit -> AMPLI_FLAG_RERUN_GENERAL, in whichever chunk of a streamed
    cohort the record sits (the table written by the earlier chunks is then void and the literal kernel redoes the cohort)."""
    rng = np.random.default_rng(11)
    P, S = 300, 12
    recs = synth_recs(P, S, depth=3000)
    s_deep = {"first": 1, "middle": 6, "last": 11}[where]
    deep = 6_000_000  # 2^22 < RD = 12e6 < 2^24
    recs[s_deep, 77] = [deep - 30, 10, 10, 10, deep - 30, 10, 10, 10]
    ref = orc.error_reduce(recs, P, 0.002, 100)
    want = orc.error_finalize(ref)
    ctx.set_record_layout("u24")
    try:
        ctx.set_reduce_compact(True)
        ctx.set_tuning(1, groups=1)
        acc = ctx.new_acc(P)
        cuts = (0, 4, 8, 12)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            rec = ctx.records(_pack(ctx, np.ascontiguousarray(recs[lo:hi]), "u24"), "u24", hi - lo)
            fin = ctx.error_reduce_records(rec, P, acc, 0.002, 100, first_sample=lo, accumulate=lo > 0, summary=True, finalize=hi == 12)
            assert ctx.last_reduce_kernel() == COMPACT["u24"]
        assert ctx.flags() & 2  # AMPLI_FLAG_RERUN_GENERAL
        ctx.set_tuning(0, general=True)
        acc2 = ctx.new_acc(P)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            rec = ctx.records(_pack(ctx, np.ascontiguousarray(recs[lo:hi]), "u24"), "u24", hi - lo)
            fin = ctx.error_reduce_records(rec, P, acc2, 0.002, 100, first_sample=lo, accumulate=lo > 0, finalize=hi == 12)
        assert ctx.flags() == 0
        assert_acc_equal(acc2, ref)
        assert_final_equal(fin, want)
    finally:
        ctx.set_tuning()
        ctx.set_record_layout("i32")
