"""GPU parity: HIP kernels (through the C ABI) against the CPU oracle on the same inputs.

Bar: bit-exact for integer planes, counts, codes, masks and for everything the
kernels compute with the reference's own operation sequence (double sums inside
the exactness envelope, fp32 rates, the text round trip); 1e-6 on p-values / Q.
"""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tests.helpers import SEED, edge_case_recs, synth_recs, synth_ref

pytestmark = pytest.mark.gpu

Q_TOL = 1e-6  # north_star: "within 1e-6 on the Poisson p-values / error rates"


def _t(x):
    import torch

    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def assert_acc_equal(acc, ref, skip=()):
    for name, plane in acc.planes().items():
        if name in skip:
            continue
        got = plane.cpu().numpy()
        exp = ref[name]
        if name in ("gm_first", "gm_first_af", "gm_rest"):
            # state only defined where a record qualified / a later record exists
            n = ref["gm_n"]
            m = n > 0 if name != "gm_rest" else n > 1
            assert np.array_equal(got[m].view(np.int32), exp[m].view(np.int32)), name
        else:
            assert np.array_equal(got.view(np.int64 if got.dtype.itemsize == 8 else np.int32),
                                  exp.view(np.int64 if exp.dtype.itemsize == 8 else np.int32)), name


def assert_final_equal(fin, ref):
    assert np.array_equal(fin.code.cpu().numpy(), ref["code"])
    assert np.array_equal(fin.rate.cpu().numpy().view(np.int32), ref["rate"].view(np.int32))
    assert np.array_equal(fin.thr.cpu().numpy().view(np.int32), ref["thr"].view(np.int32))
    assert np.array_equal(fin.germ_present.cpu().numpy(), ref["germ_present"])
    m = ref["germ_present"] > 0
    assert np.array_equal(fin.germ_val.cpu().numpy()[m].astype(np.float64), ref["germ_val"][m])


def test_synth_device_matches_host(ctx):
    for tumour in (False, True):
        d = ctx.synth_fill(777, 5, first_sample=3, depth=2000, tumour=tumour).cpu().numpy()
        h = synth_recs(777, 5, first=3, depth=2000, tumour=tumour)
        assert np.array_equal(d, h)
    assert np.array_equal(ctx.synth_ref(777).cpu().numpy(), synth_ref(777))


@pytest.mark.parametrize("groups", [0, 1, 2, 4])
@pytest.mark.parametrize("P,S,splits", [(1, 1, 0), (63, 3, 0), (64, 4, 1), (65, 5, 2), (1000, 32, 0), (1000, 33, 3),
                                        (4097, 37, 0), (10000, 32, 0), (777, 130, 0), (15, 300, 2)])
def test_error_reduce_synthetic(ctx, P, S, splits, groups):
    recs = synth_recs(P, S)
    ref = orc.error_reduce(recs, P, 0.002, 100)
    assert ref["order_sensitive"] == 0
    ctx.set_tuning(splits, groups=groups)
    acc = ctx.error_reduce(_t(recs), P, 0.002, 100)
    ctx.set_tuning(0)
    assert ctx.flags() == 0
    assert_acc_equal(acc, ref)
    fin = ctx.error_finalize(acc, 0.002, 100)
    assert_final_equal(fin, orc.error_finalize(ref))
    assert int(fin.flags.item()) == 0


@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("C,cov", [(0.002, 100), (0.01, 50), (0.0005, 1), (0.002, 1000)])
def test_error_reduce_edge_cases(ctx, C, cov, general):
    rng = np.random.default_rng(7)
    P, S = 517, 41
    recs = edge_case_recs(P, S, rng)
    ref = orc.error_reduce(recs, P, C, cov)
    ctx.set_tuning(0, general=general)
    acc = ctx.error_reduce(_t(recs), P, C, cov)
    ctx.set_tuning(0)
    assert ctx.flags() == 0
    assert_acc_equal(acc, ref)
    assert_final_equal(ctx.error_finalize(acc, C, cov), orc.error_finalize(ref))


@pytest.mark.parametrize("P,S,splits", [(65, 5, 2), (1000, 33, 3), (4097, 200, 0)])
def test_error_reduce_general_kernel_synthetic(ctx, P, S, splits):
    """The literal kernel (reference operation order, any depth) and the fast kernel agree with the oracle."""
    recs = synth_recs(P, S)
    ref = orc.error_reduce(recs, P, 0.002, 100)
    for general, groups in ((True, 1), (True, 4), (False, 2), (False, 0)):
        ctx.set_tuning(splits, general=general, groups=groups)
        acc = ctx.error_reduce(_t(recs), P, 0.002, 100)
        ctx.set_tuning(0)
        assert_acc_equal(acc, ref)


def test_fast_kernel_flags_depths_beyond_its_envelope(ctx):
    """Depths >= 2^22 are outside the fast kernel's integer envelope: it must say so, and the literal kernel
    (incl. the fp32 AF gate above 2^24 reads) must still match the oracle."""
    rng = np.random.default_rng(21)
    P, S = 200, 9
    recs = synth_recs(P, S)
    recs[3, 17] = [5_000_000, 40_000, 7, 0, 4_900_000, 38_000, 0, 9]          # > 2^22
    recs[5, 99] = [20_000_000, 900_000, 3, 1, 19_000_000, 800_000, 2, 0]      # > 2^24: float(x) is no longer exact
    recs[6, 99] = [20_000_001, 1_000_003, 3, 1, 19_000_001, 950_001, 2, 0]
    ref = orc.error_reduce(recs, P, 0.002, 100)
    ctx.flags()
    ctx.error_reduce(_t(recs), P, 0.002, 100)
    assert ctx.flags() & 2
    ctx.set_tuning(0, general=True)
    acc = ctx.error_reduce(_t(recs), P, 0.002, 100)
    ctx.set_tuning(0)
    assert ctx.flags() == 0
    assert_acc_equal(acc, ref)


def test_error_reduce_all_absent_and_empty_quorum(ctx):
    P, S = 130, 9
    recs = np.zeros((S, P, 8), np.int32)
    recs[:, :, 0] = np.iinfo(np.int32).min  # no sample has any line: n = 0 -> NaN branch (EE:1682)
    recs[:, 5] = [0, 0, 0, 500, 0, 0, 0, 500]  # one clean position
    ref = orc.error_reduce(recs, P)
    acc = ctx.error_reduce(_t(recs), P)
    assert_acc_equal(acc, ref)
    fin = ctx.error_finalize(acc)
    assert_final_equal(fin, orc.error_finalize(ref))
    code = fin.code.cpu().numpy()
    assert (code[:, 0] == 2).all() and (code[:3, 5] == 0).all()


def test_error_reduce_with_extra_occurrences(ctx):
    """Positions listed twice (or three times) per file: every line is a record."""
    rng = np.random.default_rng(11)
    P, S = 300, 13
    mult = np.zeros(P, np.int64)
    mult[rng.choice(P, 40, replace=False)] = 1
    mult[rng.choice(P, 5, replace=False)] = 2
    dup_off = np.concatenate([[0], np.cumsum(mult)]).astype(np.uint32)
    E = int(dup_off[-1])
    base = synth_recs(P, S)
    extra = edge_case_recs(E, S, rng)
    recs = np.concatenate([base, extra], axis=1)
    ref = orc.error_reduce(recs, P, 0.002, 100, E=E, dup_off=dup_off)
    for groups in (1, 2, 4):
        for general in (False, True):
            ctx.set_tuning(0, general=general, groups=groups)
            acc = ctx.error_reduce(_t(recs), P, 0.002, 100, E=E, dup_off=_t(dup_off))
            ctx.set_tuning(0)
            assert_acc_equal(acc, ref)
    assert_final_equal(ctx.error_finalize(acc), orc.error_finalize(ref))


def test_shard_merge_matches_single_pass(ctx):
    """Sample shards merged in order == one pass (the multi-GPU merge)."""
    P, S = 2000, 48
    recs = synth_recs(P, S)
    full = orc.error_reduce(recs, P)
    cuts = [0, 7, 8, 30, 48]
    parts = [ctx.error_reduce(_t(recs[a:b]), P, first_sample=a) for a, b in zip(cuts[:-1], cuts[1:])]
    merged = ctx.acc_merge(parts)
    assert_acc_equal(merged, full)
    # oracle's own merge agrees too
    o = orc.error_reduce(recs[:7], P)
    for a, b in zip(cuts[1:-1], cuts[2:]):
        o = orc.acc_merge(o, orc.error_reduce(recs[a:b], P, first_sample=a))
    for k in ("snt", "srd", "cnt", "nrec", "gm_n"):
        assert np.array_equal(o[k], full[k])
    # all-reduce style: SUM planes added, gm triples gathered and folded
    import torch

    dst = ctx.new_acc(P)
    for name in ("snt", "srd", "cnt", "nrec"):
        getattr(dst, name).copy_(sum(getattr(p, name) for p in parts))
    _, gm_off, gm_bytes = ctx.regions(P)
    gathered = torch.cat([p.buf[gm_off: gm_off + gm_bytes] for p in parts])
    ctx.gm_merge(dst, gathered, len(parts))
    assert_acc_equal(dst, full, skip=("gm_first",))  # the sample index is not part of the exchanged region


def _sliced_merge_emulated(ctx, shards, P, firsts, E=0, dup_off=None, C_value=0.002, cov=100, slim=False):
    """The position-sliced merge (SlicedMerger's data movement) with the collectives replaced by tensor ops.
    slim: the sums travel as 14 packed planes (AMPLI_SLICE_SLIM) instead of 21."""
    import torch

    from amplisolve_amd.dist import slice_geometry, slice_planes

    n = len(shards)
    L, sums_bytes, _, block_bytes = slice_geometry(P, n, slim)
    pl = slice_planes(slim)
    assert sums_bytes == n * pl * L * 8
    ctx.set_slice_format(slim)
    try:
        sums, gms = [], []
        for recs, first in zip(shards, firsts):
            s = torch.zeros(n * pl * L, dtype=torch.float64, device="cuda")
            g = torch.zeros(n * 8 * L, dtype=torch.float32, device="cuda")
            ctx.error_reduce_sliced(recs, P, n, s, g, C_value, cov, E=E, dup_off=dup_off, first_sample=first)
            sums.append(s)
            gms.append(g)
        total = torch.stack(sums).sum(0).view(n, pl * L)                      # reduce-scatter: rank k keeps row k
        blocks = torch.zeros(n * block_bytes, dtype=torch.uint8, device="cuda")
        for k in range(n):
            recv = torch.stack([g.view(n, 8 * L)[k] for g in gms]).contiguous()  # all-to-all: chunk j = rank j's pair for slice k
            ctx.error_finalize_slice(P, n, k, total[k].contiguous(), recv, blocks[k * block_bytes:(k + 1) * block_bytes], C_value, cov)
        fin = ctx.error_table_unslice(P, n, blocks)                          # all-gather: the blocks back to back
    finally:
        ctx.set_slice_format(False)
    fin.blocks = blocks
    return fin


@pytest.mark.parametrize("slim", [False, True])
@pytest.mark.parametrize("P,S,n", [(2000, 48, 2), (1000, 37, 3), (4097, 64, 8), (100, 16, 8), (15, 300, 2), (777, 130, 4)])
def test_sliced_merge_matches_single_pass(ctx, P, S, n, slim):
    """Reduce-scatter / all-to-all / all-gather merge of n sample shards == one pass over all samples, in both formats of the
    sums (21 plain planes; 14 planes with the integer planes packed, round 4)."""
    from amplisolve_amd.dist import shard_range

    recs = synth_recs(P, S)
    cuts = [shard_range(S, r, n) for r in range(n)]
    ref = orc.error_finalize(orc.error_reduce(recs, P))
    shards = [_t(recs[a:b]) for a, b in cuts if b > a]
    firsts = [a for a, b in cuts if b > a]
    if len(shards) < n:  # fewer samples than ranks cannot happen with shard_range here, but keep the emulation honest
        pytest.skip("empty shard")
    ctx.flags()
    fin = _sliced_merge_emulated(ctx, shards, P, firsts, slim=slim)
    assert ctx.flags() == 0  # no AMPLI_FLAG_SLICE_RANGE: every shard's values fit their share of the packed fields
    assert_final_equal(fin, ref)
    assert int(fin.flags.item()) == 0
    one = ctx.error_estimate(_t(recs), P)
    for a, b in ((fin.rate, one.rate), (fin.thr, one.thr), (fin.code, one.code), (fin.germ_present, one.germ_present)):
        assert bool((a == b).all()) or np.array_equal(a.cpu().numpy().view(np.uint8), b.cpu().numpy().view(np.uint8))


@pytest.mark.parametrize("P,n", [(1000, 3), (4097, 8), (100, 8)])
def test_poisson_call_reads_thresholds_from_the_gathered_blocks(ctx, P, n):
    """ampli_poisson_call_blocks (thresholds straight from the blocks of a sliced merge) == ampli_poisson_call on the
    unsliced table, both modes, and == the oracle."""
    from amplisolve_amd.api import POISSON_FULL, POISSON_PREFILTER
    from amplisolve_amd.dist import shard_range

    S, T = 40, 5
    recs = synth_recs(P, S)
    cuts = [shard_range(S, r, n) for r in range(n)]
    fin = _sliced_merge_emulated(ctx, [_t(recs[a:b]) for a, b in cuts], P, [a for a, _ in cuts])
    ref_code = synth_ref(P)
    trecs = synth_recs(P, T, tumour=True)
    exp = orc.poisson_call(trecs, P, orc.error_finalize(orc.error_reduce(recs, P))["thr"], ref_code, 100)
    for mode in (POISSON_PREFILTER, POISSON_FULL):
        a = ctx.poisson_call(_t(trecs), P, fin.thr, _t(ref_code), 100, mode=mode, capacity=1 << 16)
        b = ctx.poisson_call(_t(trecs), P, fin.blocks, _t(ref_code), 100, mode=mode, capacity=1 << 16, blocks_of=n)
        assert np.array_equal(a["call_mask"].cpu().numpy(), exp["call_mask"])
        assert np.array_equal(b["call_mask"].cpu().numpy(), exp["call_mask"])
        assert ctx.read_calls(a).tobytes() == ctx.read_calls(b).tobytes()


@pytest.mark.parametrize("slim", [False, True])
def test_sliced_merge_with_groups_of_batches(ctx, slim):
    """ampli_set_slice_group: G independent batches share one round of collectives (buffers [n][G][planes][L]); every
    batch of the group comes out as its own single pass, and poisson_call reads the right batch's blocks."""
    import torch

    from amplisolve_amd.api import POISSON_PREFILTER
    from amplisolve_amd.dist import shard_range, slice_geometry, slice_planes

    P, S, n, G, T = 1000, 24, 3, 2, 4
    L, _, _, bb = slice_geometry(P, n, slim)
    pl = slice_planes(slim)
    batches = [synth_recs(P, S, seed=900 + g) for g in range(G)]
    cuts = [shard_range(S, r, n) for r in range(n)]
    sums = [torch.zeros(n * G * pl * L, dtype=torch.float64, device="cuda") for _ in range(n)]
    gms = [torch.zeros(n * G * 8 * L, dtype=torch.float32, device="cuda") for _ in range(n)]
    ctx.set_slice_format(slim)
    try:
        for r, (a, b) in enumerate(cuts):
            for g in range(G):
                ctx.set_slice_group(G, g)
                ctx.error_reduce_sliced(_t(batches[g][a:b]), P, n, sums[r], gms[r], first_sample=a)
        total = torch.stack(sums).sum(0).view(n, G * pl * L)                       # reduce-scatter: rank k keeps chunk k
        block = [torch.zeros(G * bb, dtype=torch.uint8, device="cuda") for _ in range(n)]
        for k in range(n):
            recv = torch.stack([gm.view(n, G * 8 * L)[k] for gm in gms]).contiguous()  # all-to-all
            for g in range(G):
                ctx.set_slice_group(G, g)
                ctx.error_finalize_slice(P, n, k, total[k].contiguous(), recv, block[k])
        blocks = torch.cat(block)                                                   # all-gather: [n][G][block]
        ref_code = synth_ref(P)
        trecs = synth_recs(P, T, tumour=True)
        for g in range(G):
            ctx.set_slice_group(G, g)
            fin = ctx.error_table_unslice(P, n, blocks)
            assert_final_equal(fin, orc.error_finalize(orc.error_reduce(batches[g], P)))
            a = ctx.poisson_call(_t(trecs), P, blocks, _t(ref_code), 100, mode=POISSON_PREFILTER, capacity=1 << 16, blocks_of=n)
            b = ctx.poisson_call(_t(trecs), P, fin.thr, _t(ref_code), 100, mode=POISSON_PREFILTER, capacity=1 << 16)
            assert torch.equal(a["call_mask"], b["call_mask"])
    finally:
        ctx.set_slice_group(1, 0)
        ctx.set_slice_format(False)


def test_sliced_merge_edge_cases_and_extras(ctx):
    """Edge-case records (absent cells, depth around the cutoff, AF around 5 %, quorum failures, NaN rates) with
    positions listed more than once per file, sharded down to one sample per rank so that many shards have no
    qualifying record for a position (the -1 marker of the exchange)."""
    from amplisolve_amd.dist import shard_range

    rng = np.random.default_rng(23)
    P, S = 333, 9
    mult = np.zeros(P, np.int64)
    mult[rng.choice(P, 40, replace=False)] = 1
    mult[rng.choice(P, 5, replace=False)] = 2
    dup_off = np.concatenate([[0], np.cumsum(mult)]).astype(np.uint32)
    E = int(dup_off[-1])
    recs = edge_case_recs(P + E, S, rng)
    ref = orc.error_finalize(orc.error_reduce(recs, P, 0.002, 100, E=E, dup_off=dup_off))
    for n in (2, 3, S):
        cuts = [shard_range(S, r, n) for r in range(n)]
        for general in (False, True):
            ctx.set_tuning(0, general=general)
            fin = _sliced_merge_emulated(ctx, [_t(recs[a:b]) for a, b in cuts], P, [a for a, _ in cuts], E=E, dup_off=_t(dup_off))
            slim = _sliced_merge_emulated(ctx, [_t(recs[a:b]) for a, b in cuts], P, [a for a, _ in cuts], E=E, dup_off=_t(dup_off), slim=True)
            ctx.set_tuning(0)
            assert_final_equal(fin, ref)
            assert_final_equal(slim, ref)


def test_slim_exchange_format_flags_values_beyond_a_shards_share_of_a_field(ctx):
    """AMPLI_SLICE_SLIM packs the two strands' depth sums of a nucleotide into one double (2^26 each) and the record counts three
    to a double (2^17 each); with n shards each may use 1/n of a field.  A shard beyond its share raises AMPLI_FLAG_SLICE_RANGE
    (the caller repeats the exchange in the wide format, which has no such limit and gives the single-pass table)."""
    P, S, n = 128, 40, 8
    recs = synth_recs(P, S)
    recs[:, 7] = [300_000, 10, 5, 3, 290_000, 8, 4, 2]  # a shard of 33 such records: depth sums of 33 x 3e5 = 9.9e6 > 2^26 / 8
    ref = orc.error_finalize(orc.error_reduce(recs, P))
    ctx.flags()
    parts = [recs[:33]] + [recs[33 + i:34 + i] for i in range(7)]  # n = 8 shards, seven of them one sample
    starts = [0] + [33 + i for i in range(7)]
    fin = _sliced_merge_emulated(ctx, [_t(x) for x in parts], P, starts, slim=True)
    assert ctx.flags() & 8  # AMPLI_FLAG_SLICE_RANGE
    fin = _sliced_merge_emulated(ctx, [_t(x) for x in parts], P, starts, slim=False)
    assert ctx.flags() == 0
    assert_final_equal(fin, ref)


def test_sliced_merge_reports_the_exactness_flag(ctx):
    """A slice whose double sums leave the exactness envelope raises flag bit 0 through the gathered blocks
    (same construction as test_exactness_envelope_flag)."""
    P, S = 256, 24
    recs = synth_recs(P, S)
    ok = _sliced_merge_emulated(ctx, [_t(recs[:12]), _t(recs[12:])], P, [0, 12])
    assert int(ok.flags.item()) == 0
    bad = _sliced_merge_emulated(ctx, [_t(recs[:12]), _t(recs[12:])], P, [0, 12], C_value=1e-7, cov=1)
    assert int(bad.flags.item()) & 1


def test_text_roundtrip_device(ctx):
    import torch

    rng = np.random.default_rng(3)
    vals = np.concatenate([
        rng.random(200000, dtype=np.float32) * np.float32(0.06),
        (rng.random(100000, dtype=np.float32) * 20).astype(np.float32),
        np.float32(10.0) ** rng.uniform(-12, 3, 100000).astype(np.float32),
        np.array([0, 1e-7, 4.9e-7, 5e-7, 5.1e-7, 1.5e-6, 2.5e-6, 0.002, 0.01, 0.05, 1, 15.9999, 16, 17.5, 1e6, 1e-45, 3e38], np.float32),
        (np.arange(0, 60000, dtype=np.float64) * 1e-6 + 5e-7).astype(np.float32),  # decimal ties
    ])
    got = ctx.roundtrip_batch(torch.from_numpy(vals).cuda()).cpu().numpy()
    exp = np.array([orc.lib().oracle_text_roundtrip(float(v)) for v in vals], np.float32)
    assert np.array_equal(got.view(np.int32), exp.view(np.int32))


def test_scorer_against_oracle(ctx):
    import torch

    rng = np.random.default_rng(5)
    n = 400000
    k = rng.integers(0, 60, n).astype(np.int32)
    k[: n // 4] = rng.integers(0, 3000, n // 4)
    rd = rng.integers(0, 60000, n).astype(np.int32)
    err = rng.choice(np.array([0.002, 0.01, 0.0005, 0.05, 0.0, -1.0, 0.002189, 0.000123, 0.3], np.float32), n)
    q, p = ctx.score_batch(_t(k), _t(rd), _t(err))
    q, p = q.cpu().numpy(), p.cpu().numpy()
    qo, po = orc.score_batch(k, rd, err)
    assert np.array_equal(np.isnan(p), np.isnan(po))
    m = ~np.isnan(po)
    assert np.max(np.abs(p[m] - po[m])) <= Q_TOL  # absolute on p
    rel = np.abs(p[m] - po[m]) / np.maximum(np.abs(po[m]), 1e-300)
    big = np.abs(po[m]) > 1e-10
    assert np.max(rel[big]) <= Q_TOL
    mq = ~np.isnan(qo)
    assert np.max(np.abs(q[mq] - qo[mq])) <= 1e-5  # Q = -10 log10 p: 1e-6 relative on p is 4.3e-6 on Q
    # the gate decision (Q >= 5) must agree except within rounding of the boundary
    dis = (q[mq] >= 5) != (qo[mq] >= 5)
    assert not dis.any() or np.all(np.abs(qo[mq][dis] - 5) < 1e-9)


@pytest.mark.skipif(not __import__("os").path.exists(orc.REF_VC_SCORER), reason="oracle/_ref/libvc_scorer_ref.so (the reference's own scorer, built where /root/reference exists) is absent")
def test_device_scorer_against_the_reference_functions_directly(ctx):
    """The device scorer against mutationRulesPoissonQualityScore / kf_gammaq themselves (compiled from the reference's
    source), without the oracle in between: Q within 1e-6 relative wherever it is not clamped, the clamps and special codes
    identical, and the Q >= 5 decision of VC:898 identical on every point."""
    import ctypes as C

    R = C.CDLL(orc.REF_VC_SCORER)
    rng = np.random.default_rng(19)
    n = 300000
    k = rng.integers(0, 80, n).astype(np.int32)
    k[: n // 5] = rng.integers(0, 4000, n // 5)
    rd = rng.integers(0, 70000, n).astype(np.int32)
    err = rng.choice(np.array([0.002, 0.01, 0.0005, 0.05, 0.0, -1.0, 0.002189, 0.000123, 0.3, 0.00001], np.float32), n)
    want = np.empty(n)
    R.ref_score_batch(k.ctypes.data_as(C.c_void_p), rd.ctypes.data_as(C.c_void_p), err.ctypes.data_as(C.c_void_p), C.c_long(n),
                      want.ctypes.data_as(C.c_void_p))
    q, _ = ctx.score_batch(_t(k), _t(rd), _t(err))
    q = q.cpu().numpy()
    special = (want == -888.0) | (want == 100.0) | (want == 0.0)
    assert np.array_equal(q[special], want[special])
    m = ~special
    assert np.max(np.abs(q[m] - want[m]) / np.maximum(np.abs(want[m]), 1e-300)) <= Q_TOL
    assert np.array_equal(q >= 5, want >= 5)
    # the scorer of the all-scores mode (AMPLI_POISSON_FULL since round 4: integer-count form of the same recipe) against the
    # same reference values: specials identical, Q within 1e-6 relative, every Q >= 5 decision identical
    qd = ctx.score_dense_batch(_t(k), _t(rd), _t(err)).cpu().numpy()
    assert np.array_equal(qd[special], want[special])
    assert np.max(np.abs(qd[m] - want[m]) / np.maximum(np.abs(want[m]), 1e-300)) <= Q_TOL
    assert np.array_equal(qd >= 5, want >= 5)
    # and p itself within the north star's 1e-6 (absolute and relative): p = 10^(-Q/10) where Q is not clamped
    pw, pd = 10.0 ** (-want[m] / 10), 10.0 ** (-qd[m] / 10)
    assert np.max(np.abs(pd - pw)) <= 1e-6 and np.max(np.abs(pd - pw) / pw) <= 1e-6


@pytest.mark.parametrize("P,T", [(1, 1), (255, 3), (257, 5), (3000, 8)])
def test_poisson_call_synthetic(ctx, P, T):
    S = 24
    normals = synth_recs(P, S)
    fin = orc.error_finalize(orc.error_reduce(normals, P))
    ref_code = synth_ref(P)
    trecs = synth_recs(P, T, tumour=True)
    exp = orc.poisson_call(trecs, P, fin["thr"], ref_code, 100)
    from amplisolve_amd.api import POISSON_FULL, POISSON_PREFILTER

    full = ctx.poisson_call(_t(trecs), P, _t(fin["thr"]), _t(ref_code), 100, mode=POISSON_FULL, dense_q=True, dense_af=True,
                            capacity=4 * P * T + 64)
    assert np.array_equal(full["call_mask"].cpu().numpy(), exp["call_mask"])
    q, qo = full["q"].cpu().numpy(), exp["q"]
    assert np.array_equal(q == -1, qo == -1)
    assert np.max(np.abs(q - qo)) <= 1e-5
    assert np.array_equal(full["af"].cpu().numpy().view(np.int32), exp["af"].view(np.int32))
    pre = ctx.poisson_call(_t(trecs), P, _t(fin["thr"]), _t(ref_code), 100, mode=POISSON_PREFILTER, capacity=4 * P * T + 64)
    assert np.array_equal(pre["call_mask"].cpu().numpy(), exp["call_mask"])
    # compact list == mask, with the oracle's Q and VAFs
    for res in (full, pre):
        calls = ctx.read_calls(res)
        t_i, r_i = np.nonzero(exp["call_mask"])
        n_exp = sum(bin(int(v)).count("1") for v in exp["call_mask"][t_i, r_i])
        assert len(calls) == n_exp
        for c in calls:
            assert exp["call_mask"][c["sample"], c["record"]] >> c["alt"] & 1
            assert abs(c["q_fw"] - qo[c["sample"], c["record"], c["alt"], 0]) <= 1e-5
            assert abs(c["q_bw"] - qo[c["sample"], c["record"], c["alt"], 1]) <= 1e-5
            assert np.float32(c["af"]) == exp["af"][c["sample"], c["record"], c["alt"], 0]


def test_poisson_call_edge_cases(ctx):
    """Special thresholds (-1 -> Q=-888, 0 -> magic 0.0010008), non-ACGT reference, absent records, extras."""
    rng = np.random.default_rng(13)
    P, T, E = 400, 6, 37
    trecs = edge_case_recs(P + E, T, rng)
    # lift some cells to strong variants so calls exist
    trecs[:, ::7, :] = np.array([30, 0, 0, 400, 25, 0, 0, 380], np.int32)
    thr = rng.choice(np.array([0.002, 0.01, 0.0, -1.0, 0.000731, 0.05], np.float32), size=(2, 4, P)).astype(np.float32)
    ref_code = rng.integers(0, 4, P).astype(np.uint8)
    ref_code[::11] = 255
    ext_pos = rng.integers(0, P, E).astype(np.uint32)
    exp = orc.poisson_call(trecs, P, thr, ref_code, 100, E=E, ext_pos=ext_pos)
    from amplisolve_amd.api import POISSON_FULL, POISSON_PREFILTER

    for mode in (POISSON_FULL, POISSON_PREFILTER):
        res = ctx.poisson_call(_t(trecs), P, _t(thr), _t(ref_code), 100, mode=mode, E=E, ext_pos=_t(ext_pos),
                               dense_q=(mode == POISSON_FULL))
        assert np.array_equal(res["call_mask"].cpu().numpy(), exp["call_mask"])
        if mode == POISSON_FULL:
            q = res["q"].cpu().numpy()
            assert np.array_equal(q == -1, exp["q"] == -1) and np.max(np.abs(q - exp["q"])) <= 1e-5
    assert exp["call_mask"].any()


def test_full_size_properties(ctx):
    """BASELINE config 3 (100k x 256 normals x 96 tumours) through size-independent properties."""
    import torch

    from amplisolve_amd.api import POISSON_FULL, POISSON_PREFILTER

    P, S, T = 100_000, 256, 96
    recs = ctx.synth_fill(P, S)
    acc = ctx.error_reduce(recs, P)
    # (1) shard invariance: two sample shards merged in order == one pass, bit for bit
    a = ctx.error_reduce(recs[:100], P, first_sample=0)
    b = ctx.error_reduce(recs[100:], P, first_sample=100)
    m = ctx.acc_merge([a, b])
    for name, plane in acc.planes().items():
        other = m.planes()[name]
        if name in ("gm_first", "gm_first_af"):
            sel = acc.gm_n > 0
            assert torch.equal(plane[sel], other[sel]), name
        elif name == "gm_rest":
            sel = acc.gm_n > 1
            assert torch.equal(plane[sel], other[sel]), name
        else:
            assert torch.equal(plane, other), name
    # (2) conservation: nrec == number of present records; srd sums bounded by total depth
    present = (recs[:, :, 0] != torch.iinfo(torch.int32).min)
    assert torch.equal(acc.nrec.long(), present.sum(0))
    assert bool((acc.cnt <= acc.nrec.unsqueeze(0)).all())
    # (3) idempotence
    acc2 = ctx.error_reduce(recs, P)
    for name, plane in acc.planes().items():  # (the padding between planes is not written)
        assert torch.equal(plane, acc2.planes()[name]), name
    # (4) oracle on a slice of positions (records are independent across positions)
    sl = slice(31_000, 31_900)
    ref = orc.error_reduce(recs[:, sl].cpu().numpy(), 900)
    assert np.array_equal(acc.snt[:, :, sl].cpu().numpy(), ref["snt"])
    assert np.array_equal(acc.srd[:, :, sl].cpu().numpy(), ref["srd"])
    assert np.array_equal(acc.cnt[:, sl].cpu().numpy(), ref["cnt"])
    fin = ctx.error_finalize(acc)
    assert int(fin.flags.item()) == 0
    reff = orc.error_finalize(ref)
    assert np.array_equal(fin.thr[:, :, sl].cpu().numpy().view(np.int32), reff["thr"].view(np.int32))
    # (5) calling: prefilter mode == full mode, and the oracle on a slice
    trecs = ctx.synth_fill(P, T, tumour=True)
    refc = ctx.synth_ref(P)
    full = ctx.poisson_call(trecs, P, fin.thr, refc, 100, mode=POISSON_FULL)
    pre = ctx.poisson_call(trecs, P, fin.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 20)
    assert torch.equal(full["call_mask"], pre["call_mask"])
    n_calls = ctx.n_calls_total(pre)
    bits = sum(int(((pre["call_mask"] >> a) & 1).sum()) for a in range(4))
    assert n_calls == bits and n_calls > 0
    exp = orc.poisson_call(trecs[:4, sl].cpu().numpy(), 900, fin.thr[:, :, sl].cpu().numpy(), refc[sl].cpu().numpy(), 100, dense=False)
    assert np.array_equal(pre["call_mask"][:4, sl].cpu().numpy(), exp["call_mask"])


def test_deep_coverage_stress_config5_shape(ctx):
    """BASELINE config 5 shape at reduced P: depth 50 000x, 256 normals.  Sums are ~25x larger than at config 3:
    the exactness envelope must still hold (flag 0) and a slice must match the oracle bit for bit."""
    import torch

    P, S, T, depth = 150_000, 256, 8, 50_000
    recs = ctx.synth_fill(P, S, depth=depth)
    acc = ctx.error_reduce(recs, P)
    assert ctx.flags() == 0
    fin = ctx.error_finalize(acc)
    assert int(fin.flags.item()) == 0
    sl = slice(77_000, 77_700)
    sub = recs[:, sl].cpu().numpy()
    ref = orc.error_reduce(sub, 700)
    assert ref["order_sensitive"] == 0
    assert np.array_equal(acc.snt[:, :, sl].cpu().numpy(), ref["snt"])
    assert np.array_equal(acc.srd[:, :, sl].cpu().numpy(), ref["srd"])
    assert np.array_equal(acc.cnt[:, sl].cpu().numpy(), ref["cnt"])
    reff = orc.error_finalize(ref)
    assert np.array_equal(fin.thr[:, :, sl].cpu().numpy().view(np.int32), reff["thr"].view(np.int32))
    m = reff["germ_present"] > 0
    assert np.array_equal(fin.germ_val[:, sl].cpu().numpy()[m].astype(np.float64), reff["germ_val"][m])
    trecs = ctx.synth_fill(P, T, depth=depth, tumour=True)
    refc = ctx.synth_ref(P)
    from amplisolve_amd.api import POISSON_FULL, POISSON_PREFILTER

    pre = ctx.poisson_call(trecs, P, fin.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 20)
    assert ctx.flags() == 0
    full = ctx.poisson_call(trecs, P, fin.thr, refc, 100, mode=POISSON_FULL)
    assert torch.equal(pre["call_mask"], full["call_mask"])
    exp = orc.poisson_call(trecs[:, sl].cpu().numpy(), 700, fin.thr[:, :, sl].cpu().numpy(), refc[sl].cpu().numpy(), 100, dense=False)
    assert np.array_equal(pre["call_mask"][:, sl].cpu().numpy(), exp["call_mask"])
    assert ctx.n_calls_total(pre) > 0


@pytest.mark.parametrize("P,S,splits", [(64, 4, 1), (1000, 33, 3), (5000, 64, 0), (33, 200, 0)])
def test_error_estimate_fused_equals_two_step(ctx, P, S, splits):
    """ampli_error_estimate (finalize fused into the reduce epilogue, table optional) == reduce + finalize."""
    import torch

    recs = _t(synth_recs(P, S))
    ref = orc.error_finalize(orc.error_reduce(recs.cpu().numpy(), P, 0.002, 100))
    ctx.set_tuning(splits)
    fused = ctx.error_estimate(recs, P, 0.002, 100)                      # no table at all
    acc = ctx.new_acc(P)
    fused2 = ctx.error_estimate(recs, P, 0.002, 100, acc=acc)            # table as a by-product
    two = ctx.error_finalize(ctx.error_reduce(recs, P, 0.002, 100), 0.002, 100)
    ctx.set_tuning(0)
    for f in (fused, fused2, two):
        assert_final_equal(f, ref)
    ref_acc = orc.error_reduce(recs.cpu().numpy(), P, 0.002, 100)
    assert_acc_equal(acc, ref_acc)


def test_prefilter_queue_overflow_is_flagged_and_recoverable(ctx):
    """Every record carries strong variants on both strands: far more survivors than the default queue (T*R/4 items)
    holds.  The kernel must raise AMPLI_FLAG_QUEUE_OVERFLOW rather than silently drop work; with a larger queue the
    result equals the oracle and the full-mode kernel."""
    import torch

    from amplisolve_amd.api import POISSON_FULL, POISSON_PREFILTER

    P, T = 70_000, 4
    rng = np.random.default_rng(17)
    trecs = np.zeros((T, P, 8), np.int32)
    trecs[:, :, 0] = 900; trecs[:, :, 4] = 850                      # reference A
    trecs[:, :, 1] = rng.integers(20, 60, (T, P)); trecs[:, :, 5] = rng.integers(20, 60, (T, P))
    trecs[:, :, 2] = rng.integers(15, 50, (T, P)); trecs[:, :, 6] = rng.integers(15, 50, (T, P))
    trecs[:, :, 3] = rng.integers(10, 40, (T, P)); trecs[:, :, 7] = rng.integers(10, 40, (T, P))
    thr = np.full((2, 4, P), 0.002, np.float32)
    ref_code = np.zeros(P, np.uint8)
    d_t, d_thr, d_ref = _t(trecs), _t(thr), _t(ref_code)
    from amplisolve_amd import Context

    ctx = Context(0)  # a fresh context: its queue has the default size, not what earlier tests grew it to
    ctx.poisson_call(d_t, P, d_thr, d_ref, 100, mode=POISSON_PREFILTER)
    assert ctx.flags() & 4
    ctx.set_queue_items(3 * T * P)
    pre = ctx.poisson_call(d_t, P, d_thr, d_ref, 100, mode=POISSON_PREFILTER, capacity=4 * T * P)  # segments fill unevenly: leave slack
    assert ctx.flags() == 0
    ctx.set_queue_items(0)
    full = ctx.poisson_call(d_t, P, d_thr, d_ref, 100, mode=POISSON_FULL)
    assert torch.equal(pre["call_mask"], full["call_mask"])
    exp = orc.poisson_call(trecs[:, :2000], 2000, thr[:, :, :2000], ref_code[:2000], 100, dense=False)
    assert np.array_equal(pre["call_mask"][:, :2000].cpu().numpy(), exp["call_mask"])
    assert ctx.n_calls_total(pre) == 3 * T * P  # every alt of every record is a call
    calls = ctx.read_calls(pre)
    assert len(calls) == 3 * T * P and (calls["q_fw"] >= 5).all() and (calls["q_bw"] >= 5).all()
    ctx.close()


def test_async_drain_equals_sync(ctx):
    """The drain kernel on a side stream (opt-in) with other work enqueued behind it gives the same mask and calls."""
    import torch

    from amplisolve_amd.api import POISSON_PREFILTER

    P, S, T = 30_000, 32, 12
    normals = ctx.synth_fill(P, S)
    tum = ctx.synth_fill(P, T, tumour=True)
    refc = ctx.synth_ref(P)
    fin = ctx.error_estimate(normals, P)
    sync = ctx.poisson_call(tum, P, fin.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 18)
    sync_calls = ctx.read_calls(sync)
    ctx.set_async_drain(True)
    try:
        for _ in range(3):
            res = ctx.poisson_call(tum, P, fin.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 18)
            other = ctx.error_estimate(normals, P)      # unrelated work right behind it, into other buffers
        ctx.wait_calls()
        assert torch.equal(res["call_mask"], sync["call_mask"])
        calls = ctx.read_calls(res)
        assert len(calls) == len(sync_calls) > 0
        for f in ("sample", "record", "alt", "q_fw", "q_bw", "af"):
            assert np.array_equal(calls[f], sync_calls[f])
        assert torch.equal(other.thr, fin.thr)
    finally:
        ctx.set_async_drain(False)
    assert ctx.flags() == 0


@pytest.mark.parametrize("shards", [1, 2, 4, 8])
def test_shard_count_invariance(ctx, shards):
    """1 / 2 / 4 / 8 emulated ranks on one device (contiguous sample shards, packed SUM + ordered gm fold, exactly the
    collectives' arithmetic) give the single-pass table and therefore the same error table, bit for bit."""
    import torch

    from amplisolve_amd.dist import shard_range

    P, S = 3000, 67
    recs = synth_recs(P, S)
    full = orc.error_reduce(recs, P)
    d_recs = _t(recs)
    parts = []
    for r in range(shards):
        lo, hi = shard_range(S, r, shards)
        parts.append(ctx.error_reduce(d_recs[lo:hi].contiguous(), P, first_sample=lo))
    # what the all-reduce computes: the sum of the packed buffers; what the all-gather delivers: the gm regions in rank order
    packed = [torch.empty(21 * P, dtype=torch.float64, device="cuda") for _ in parts]
    for a, pk in zip(parts, packed):
        ctx.acc_pack(a, pk)
    total = torch.stack(packed).sum(0)
    _, gm_off, gm_bytes = ctx.regions(P)
    gathered = torch.cat([a.buf[gm_off: gm_off + gm_bytes] for a in parts])
    dst = ctx.new_acc(P)
    dst.buf.zero_()
    ctx.acc_unpack(total, dst)
    ctx.gm_merge(dst, gathered, shards)
    assert_acc_equal(dst, full, skip=("gm_first",))
    assert_final_equal(ctx.error_finalize(dst), orc.error_finalize(full))


@pytest.mark.parametrize("shards,S", [(2, 40), (4, 67), (8, 300)])
def test_packed_shard_path_equals_single_pass(ctx, shards, S):
    """The multi-GPU fast path: error_reduce_packed per shard, SUM of the packed buffers (= the all-reduce), gm regions
    in rank order (= the all-gather), error_finalize_merged  ==  the oracle's single pass.  S = 300 also exercises the
    internal sample split (partial tables -> merge -> pack)."""
    import torch

    from amplisolve_amd.dist import shard_range

    P = 2500
    recs = synth_recs(P, S)
    ref = orc.error_finalize(orc.error_reduce(recs, P))
    d_recs = _t(recs)
    _, gm_off, gm_bytes = ctx.regions(P)
    packed, regions = [], []
    for r in range(shards):
        lo, hi = shard_range(S, r, shards)
        acc = ctx.new_acc(P)
        pk = torch.empty(21 * P, dtype=torch.float64, device="cuda")
        ctx.error_reduce_packed(d_recs[lo:hi].contiguous(), P, acc, pk, first_sample=lo)
        packed.append(pk)
        regions.append(acc.buf[gm_off: gm_off + gm_bytes].clone())
    fin = ctx.error_finalize_merged(P, torch.stack(packed).sum(0), torch.cat(regions), shards)
    assert_final_equal(fin, ref)
    assert ctx.flags() == 0


def test_hipgraph_capture_replays_the_same_pass():
    """A whole pass (error_estimate -> poisson_call) captured into a hipGraph on the context's own stream and
    replayed gives the same table, mask and calls as the eager pass (small, launch-bound panel)."""
    import torch

    from amplisolve_amd import Context
    from amplisolve_amd.api import POISSON_PREFILTER

    g_ctx = Context(0, own_stream=True)
    P, S, T = 10_000, 32, 8
    normals = g_ctx.synth_fill(P, S)
    tum = g_ctx.synth_fill(P, T, tumour=True)
    refc = g_ctx.synth_ref(P)
    g_ctx.sync()
    fin = g_ctx.error_estimate(normals, P)
    res = g_ctx.poisson_call(tum, P, fin.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 16)
    g_ctx.sync()
    eager_mask = res["call_mask"].clone()
    eager_thr = fin.thr.clone()
    eager_calls = g_ctx.read_calls(res)
    torch.cuda.synchronize()

    def one_pass():
        g_ctx.error_estimate(normals, P, out=fin)
        g_ctx.poisson_call(tum, P, fin.thr, refc, 100, mode=POISSON_PREFILTER, call_mask=res["call_mask"], capacity=res["capacity"],
                           calls_buf=res["calls_buf"], n_calls=res["n_calls"])

    g_ctx.graph_begin()
    one_pass()
    graph = g_ctx.graph_end()
    res["call_mask"].zero_(); fin.thr.zero_()
    torch.cuda.synchronize()
    for _ in range(3):
        g_ctx.graph_launch(graph)
    g_ctx.sync()
    assert torch.equal(res["call_mask"], eager_mask) and torch.equal(fin.thr, eager_thr)
    calls = g_ctx.read_calls(res)
    assert len(calls) == len(eager_calls) > 0 and np.array_equal(calls["record"], eager_calls["record"])
    # (how long a replay takes next to the eager enqueue is a measurement, not a correctness property: tools/graph_timing.py)
    g_ctx.graph_destroy(graph)
    g_ctx.close()


@pytest.mark.parametrize("name,P,S,T,depth", [("config4", 100_000, 1024, 1024, 2000), ("config5", 1_000_000, 256, 64, 50_000)])
def test_baseline_configs_4_and_5_full_size(ctx, name, P, S, T, depth):
    """BASELINE configs 4 and 5 at their FULL sizes on one GPU (6.6 GB / 10 GB of records): exactness flags clear,
    a slice of the error table and of the call masks equal to the oracle, prefilter == conservation properties."""
    import torch

    from amplisolve_amd.api import POISSON_PREFILTER

    nor = ctx.synth_fill(P, S, depth=depth)
    tum = ctx.synth_fill(P, T, depth=depth, tumour=True)
    refc = ctx.synth_ref(P)
    fin = ctx.error_estimate(nor, P)
    res = ctx.poisson_call(tum, P, fin.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 22)
    assert ctx.flags() == 0 and int(fin.flags.item()) == 0
    sl = slice(P // 3, P // 3 + 300)
    o_acc = orc.error_reduce(nor[:, sl].cpu().numpy(), 300)
    assert o_acc["order_sensitive"] == 0
    o_fin = orc.error_finalize(o_acc)
    assert np.array_equal(fin.thr[:, :, sl].cpu().numpy().view(np.int32), o_fin["thr"].view(np.int32))
    assert np.array_equal(fin.code[:, sl].cpu().numpy(), o_fin["code"])
    o_call = orc.poisson_call(tum[:8, sl].cpu().numpy(), 300, o_fin["thr"], refc[sl].cpu().numpy(), 100, dense=False)
    assert np.array_equal(res["call_mask"][:8, sl].cpu().numpy(), o_call["call_mask"])
    n = ctx.n_calls_total(res)
    bits = sum(int(((res["call_mask"] >> a) & 1).sum()) for a in range(4))
    assert n == bits > 0
    # the packed 24-byte layout (what bench.py keeps resident) at the same full size: every output identical
    from amplisolve_amd import Context

    c24 = Context(0)
    c24.set_record_layout("u24")
    n24, fits_n = c24.pack24(nor)
    t24, fits_t = c24.pack24(tum)
    assert fits_n and fits_t
    fin24 = c24.error_estimate(n24, P)
    res24 = c24.poisson_call(t24, P, fin24.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 22)
    assert c24.flags() == 0 and int(fin24.flags.item()) == 0
    for k in ("rate", "thr", "code", "germ_present"):
        assert torch.equal(getattr(fin, k).view(torch.uint8), getattr(fin24, k).view(torch.uint8)), k
    assert torch.equal(res["call_mask"], res24["call_mask"]) and c24.n_calls_total(res24) == n
    c24.close()
    del n24, t24, fin24, res24
    if depth <= 2000:
        # config 4 in the uint16 layout the products hold it in: the compact-state kernel at S = 1024 (256 rows per wave), whole, as
        # four position ranges inside the library, and streamed in four chunks of 256 samples through the accumulator table -- every
        # output identical to the int32 pass above (whose slice the oracle checked)
        c16 = Context(0)
        c16.set_record_layout("u16")
        n16, fits_n = c16.pack16(nor)
        t16, fits_t = c16.pack16(tum)
        assert fits_n and fits_t
        fin16 = c16.error_estimate(n16, P)
        assert c16.last_reduce_kernel() == "error_reduce_u16_kernel"
        res16 = c16.poisson_call(t16, P, fin16.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 22)
        assert c16.flags() == 0 and int(fin16.flags.item()) == 0
        for k in ("rate", "thr", "code", "germ_present"):
            assert torch.equal(getattr(fin, k).view(torch.uint8), getattr(fin16, k).view(torch.uint8)), k
        assert torch.equal(res["call_mask"], res16["call_mask"]) and c16.n_calls_total(res16) == n
        c16.set_ranges(4)
        finr = c16.error_estimate(n16, P)
        resr = c16.poisson_call(t16, P, finr.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 22)
        assert c16.flags() == 0  # joins
        c16.set_ranges(1)
        for k in ("rate", "thr", "code", "germ_present"):
            assert torch.equal(getattr(fin, k).view(torch.uint8), getattr(finr, k).view(torch.uint8)), k
        assert torch.equal(res["call_mask"], resr["call_mask"]) and c16.n_calls_total(resr) == n
        acc = c16.new_acc(P)
        v16 = n16.view(S, P, 8)
        fins = None
        for ci in range(4):
            rec = c16.records(v16[ci * 256:(ci + 1) * 256], "u16", 256)
            fins = c16.error_reduce_records(rec, P, acc, 0.002, 100, first_sample=ci * 256, accumulate=ci > 0, finalize=ci == 3, summary=True)
            assert c16.last_reduce_kernel() == "error_reduce_u16_kernel"
        assert c16.flags() == 0
        for k in ("rate", "thr", "code", "germ_present"):
            assert torch.equal(getattr(fin, k).view(torch.uint8), getattr(fins, k).view(torch.uint8)), k
        c16.close()
        del n16, t16, fin16, res16, finr, resr, acc, fins
    del nor, tum, res, fin
    torch.cuda.empty_cache()


def test_exactness_envelope_flag(ctx):
    """With a tiny C and coverage cutoff 1 the fp32 products carry bits far below the running sums: the double
    accumulation is then no longer exact (hence order dependent) and error_finalize must say so (flag bit 0) instead of
    passing an order-dependent table on; the usual parameters keep the flag clear on the same records."""
    P, S = 256, 24
    recs = synth_recs(P, S)
    d = _t(recs)
    ok = ctx.error_estimate(d, P, 0.002, 100)
    assert int(ok.flags.item()) == 0
    bad = ctx.error_estimate(d, P, 1e-7, 1)
    assert int(bad.flags.item()) & 1
    two = ctx.error_finalize(ctx.error_reduce(d, P, 1e-7, 1), 1e-7, 1)
    assert int(two.flags.item()) & 1
