"""The compact record layouts (AMPLI_RECORDS_U16: 8 x uint16 = 16 B per record; AMPLI_RECORDS_U24: 8 x 24 bits = 24 B)
give the same results as the oracle on the same records, through every kernel that reads records."""
import numpy as np
import pytest

from oracle import pyoracle as orc
from tests.helpers import edge_case_recs, synth_recs, synth_ref
from tests.test_gpu_parity import _t, assert_acc_equal, assert_final_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["u16", "u24"])
def ctx16(request):
    """A context on one of the compact layouts (the name dates from the first of them)."""
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from amplisolve_amd import Context

    c = Context(0)
    c.set_record_layout(request.param)
    c.layout_name = request.param
    yield c
    c.close()


def host_pack(recs, layout):
    """numpy twin of ampli_records_pack16 / ampli_records_pack24"""
    absent = recs[..., 0] == np.iinfo(np.int32).min
    v = recs.astype(np.int64)
    if layout == "u16":
        v[absent, 0] = 0xFFFF
        return v.astype(np.uint16).view(np.uint8).reshape(recs.shape[:-1] + (16,))
    v[absent, 0] = 0xFFFFFF
    b = np.stack([(v >> s) & 0xFF for s in (0, 8, 16)], axis=-1).astype(np.uint8)  # [..., 8 fields, 3 bytes] little-endian
    return b.reshape(recs.shape[:-1] + (24,))


def to16(ctx16, recs):
    """int32 numpy records -> device records in the context's layout through ampli_records_pack16/24, cross-checked
    against the host packing."""
    packed, fits = ctx16.pack(_t(recs), ctx16.layout_name)
    assert fits
    assert np.array_equal(packed.cpu().numpy().view(np.uint8).reshape(recs.shape[:-1] + (-1,)), host_pack(recs, ctx16.layout_name))
    return packed


def test_pack_flags_counts_that_do_not_fit(ctx16):
    top = 65534 if ctx16.layout_name == "u16" else (1 << 24) - 2
    recs = synth_recs(100, 3)
    recs[1, 7, 2] = top
    _, fits = ctx16.pack(_t(recs), ctx16.layout_name)
    assert fits
    recs[2, 50, 5] = top + 1  # the absent marker lives at this value in field 0: not allowed anywhere
    _, fits = ctx16.pack(_t(recs), ctx16.layout_name)
    assert not fits


@pytest.mark.parametrize("general", [False, True])
@pytest.mark.parametrize("P,S,splits,groups", [(1, 1, 0, 0), (63, 3, 0, 1), (65, 5, 2, 2), (1000, 33, 3, 4), (4097, 37, 0, 0), (777, 130, 0, 0),
                                               (15, 300, 2, 0)])
def test_error_reduce_u16(ctx16, P, S, splits, groups, general):
    recs = synth_recs(P, S)
    ref = orc.error_reduce(recs, P, 0.002, 100)
    ctx16.set_tuning(splits, general=general, groups=groups)
    acc = ctx16.error_reduce(to16(ctx16, recs), P, 0.002, 100)
    ctx16.set_tuning(0)
    assert ctx16.flags() == 0
    assert_acc_equal(acc, ref)
    assert_final_equal(ctx16.error_finalize(acc, 0.002, 100), orc.error_finalize(ref))
    assert_final_equal(ctx16.error_estimate(to16(ctx16, recs), P, 0.002, 100), orc.error_finalize(ref))


@pytest.mark.parametrize("C,cov", [(0.002, 100), (0.0005, 1), (0.002, 1000)])
def test_error_reduce_u16_edge_cases_with_extras(ctx16, C, cov):
    rng = np.random.default_rng(31)
    P, S = 300, 13
    mult = np.zeros(P, np.int64)
    mult[rng.choice(P, 40, replace=False)] = 1
    mult[rng.choice(P, 5, replace=False)] = 2
    dup_off = np.concatenate([[0], np.cumsum(mult)]).astype(np.uint32)
    E = int(dup_off[-1])
    recs = edge_case_recs(P + E, S, rng)  # depths up to 33395 per strand: fits
    ref = orc.error_reduce(recs, P, C, cov, E=E, dup_off=dup_off)
    for general in (False, True):
        ctx16.set_tuning(0, general=general)
        acc = ctx16.error_reduce(to16(ctx16, recs), P, C, cov, E=E, dup_off=_t(dup_off))
        ctx16.set_tuning(0)
        assert_acc_equal(acc, ref)
    assert_final_equal(ctx16.error_finalize(acc, C, cov), orc.error_finalize(ref))


def test_poisson_call_u16(ctx16):
    from amplisolve_amd.api import POISSON_FULL, POISSON_PREFILTER

    rng = np.random.default_rng(13)
    P, T, E = 400, 6, 37
    trecs = edge_case_recs(P + E, T, rng)
    trecs[:, ::7, :] = np.array([30, 0, 0, 400, 25, 0, 0, 380], np.int32)
    thr = rng.choice(np.array([0.002, 0.01, 0.0, -1.0, 0.000731, 0.05], np.float32), size=(2, 4, P)).astype(np.float32)
    ref_code = rng.integers(0, 4, P).astype(np.uint8)
    ref_code[::11] = 255
    ext_pos = rng.integers(0, P, E).astype(np.uint32)
    exp = orc.poisson_call(trecs, P, thr, ref_code, 100, E=E, ext_pos=ext_pos)
    assert exp["call_mask"].any()
    t16 = to16(ctx16, trecs)
    for mode in (POISSON_FULL, POISSON_PREFILTER):
        res = ctx16.poisson_call(t16, P, _t(thr), _t(ref_code), 100, mode=mode, E=E, ext_pos=_t(ext_pos), dense_q=(mode == POISSON_FULL),
                                 dense_af=(mode == POISSON_FULL), capacity=1 << 17)
        assert np.array_equal(res["call_mask"].cpu().numpy(), exp["call_mask"])
        if mode == POISSON_FULL:
            q = res["q"].cpu().numpy()
            assert np.array_equal(q == -1, exp["q"] == -1) and np.max(np.abs(q - exp["q"])) <= 1e-5
            assert np.array_equal(res["af"].cpu().numpy().view(np.int32), exp["af"].view(np.int32))
        calls = ctx16.read_calls(res)
        assert len(calls) == sum(bin(int(v)).count("1") for v in exp["call_mask"].ravel())
        for c in calls:
            assert exp["call_mask"][c["sample"], c["record"]] >> c["alt"] & 1
            assert abs(c["q_fw"] - exp["q"][c["sample"], c["record"], c["alt"], 0]) <= 1e-5
            assert np.float32(c["af"]) == exp["af"][c["sample"], c["record"], c["alt"], 0]


def test_sliced_merge_u16(ctx16):
    import torch

    from amplisolve_amd.dist import shard_range, slice_geometry

    P, S, n = 1000, 37, 3
    recs = synth_recs(P, S)
    ref = orc.error_finalize(orc.error_reduce(recs, P))
    L, _, _, bb = slice_geometry(P, n)
    sums, gms = [], []
    for r in range(n):
        a, b = shard_range(S, r, n)
        s = torch.zeros(n * 21 * L, dtype=torch.float64, device="cuda")
        g = torch.zeros(n * 8 * L, dtype=torch.float32, device="cuda")
        ctx16.error_reduce_sliced(to16(ctx16, recs[a:b]), P, n, s, g, first_sample=a)
        sums.append(s)
        gms.append(g)
    total = torch.stack(sums).sum(0).view(n, 21 * L)
    blocks = torch.zeros(n * bb, dtype=torch.uint8, device="cuda")
    for k in range(n):
        recv = torch.stack([g.view(n, 8 * L)[k] for g in gms]).contiguous()
        ctx16.error_finalize_slice(P, n, k, total[k].contiguous(), recv, blocks[k * bb:(k + 1) * bb])
    assert_final_equal(ctx16.error_table_unslice(P, n, blocks), ref)


def test_config3_full_size_u16_equals_i32(ctx, ctx16):
    """BASELINE config 3 at full size: both layouts, every output plane, the call mask and the call list."""
    import torch

    P, S, T = 100_000, 256, 96
    normals = ctx.synth_fill(P, S, first_sample=0, depth=2000)
    tumours = ctx.synth_fill(P, T, first_sample=0, depth=2000, tumour=True)
    ref_code = ctx.synth_ref(P)
    a = ctx.error_estimate(normals, P)
    n16, fits = ctx16.pack(normals, ctx16.layout_name)
    assert fits
    b = ctx16.error_estimate(n16, P)
    for k in ("rate", "thr", "code", "germ_present"):
        assert torch.equal(getattr(a, k).view(torch.uint8), getattr(b, k).view(torch.uint8)), k
    m = a.germ_present > 0
    assert torch.equal(a.germ_val[m], b.germ_val[m])
    t16, fits = ctx16.pack(tumours, ctx16.layout_name)
    assert fits
    ra = ctx.poisson_call(tumours, P, a.thr, ref_code, 100, capacity=1 << 20)
    rb = ctx16.poisson_call(t16, P, a.thr, ref_code, 100, capacity=1 << 20)
    assert torch.equal(ra["call_mask"], rb["call_mask"]) and int(ra["call_mask"].count_nonzero()) > 1000
    ca, cb = ctx.read_calls(ra), ctx16.read_calls(rb)
    assert len(ca) > 1000 and ca.tobytes() == cb.tobytes()


def test_error_reduce_u16_largest_depths(ctx16):
    """Counts at the top of the uint16 range (total depths of 350-400 k): sums, gates and the Germ_Max products stay exact."""
    P, S = 640, 40
    recs = synth_recs(P, S)
    rng = np.random.default_rng(77)
    for s_i, p_i in zip(rng.integers(2, S, 60), rng.integers(0, P, 60)):
        t = int(rng.integers(500, 6000))
        recs[s_i, p_i] = [60000, 60000, 60000, t, 59000, 61000, 60500, t + 7]      # RD ~ 366 k, T allele at 0.3 - 3 %
    recs[1, 5] = [65534, 65534, 65534, 3000, 65534, 65534, 65534, 3100]             # RD = 399 304 in the second row of a chunk
    recs[S - 1, 5] = [2000, 3, 1, 60, 1900, 2, 0, 55]                               # a later, ordinary record must still compare right
    if ctx16.layout_name == "u24":  # and depths of millions, still inside the fast kernel's 2^22 envelope
        recs[3, 9] = [1_900_000, 80_000, 7, 0, 1_800_000, 75_000, 0, 9]
        recs[7, 9] = [1_000_000, 30_000, 3, 1, 990_000, 29_000, 2, 0]
    ref = orc.error_reduce(recs, P, 0.002, 100)
    for general, groups in ((False, 1), (False, 2), (True, 1)):
        ctx16.set_tuning(0, general=general, groups=groups)
        acc = ctx16.error_reduce(to16(ctx16, recs), P, 0.002, 100)
        ctx16.set_tuning(0)
        assert ctx16.flags() == 0
        assert_acc_equal(acc, ref)
    assert_final_equal(ctx16.error_estimate(to16(ctx16, recs), P, 0.002, 100), orc.error_finalize(ref))


@pytest.mark.parametrize("P,S", [(1, 1), (63, 2), (64, 3), (65, 4), (129, 5), (200, 6), (100, 7), (1000, 33), (4097, 37), (777, 130), (300, 700),
                                 (20000, 256), (130, 4096)])
def test_error_estimate_compact_state_kernel(ctx16, P, S):
    """error_reduce_u16_kernel (round 4: uint16 records, compact per-position state, five waves per SIMD; the default for the
    launches it covers) against the general kernel and the oracle: the same table, bit for bit -- synthetic cohorts with edge-case
    records mixed in (absent cells, depth around the cutoff, AF around 5 %, empty strands, first qualifying records late in a chunk)."""
    recs = synth_recs(P, S)
    rng = np.random.default_rng(P * 1000 + S)
    if P >= 63:
        e = edge_case_recs(P, S, rng)
        pick = rng.random((S, P)) < 0.3
        recs[pick] = e[pick]
    want = orc.error_finalize(orc.error_reduce(recs, P, 0.002, 100))
    packed = to16(ctx16, recs)
    outs = []
    for compact in (True, False):
        ctx16.set_reduce_compact(compact)
        # one sample split, one lane group: the shape the compact kernel takes (left to itself the library cuts panels this small
        # along lanes or samples and they go through the general kernel)
        ctx16.set_tuning(1 if compact else 0, groups=1 if compact else 0)
        try:
            got = ctx16.error_estimate(packed, P, 0.002, 100)
            assert ctx16.flags() == 0
            fits = ctx16.layout_name == "u16" or S <= 4 * 1023  # the 24-bit form keeps 32-bit depth sums: at most 1023 records per lane
            assert ctx16.last_reduce_kernel() == (f"error_reduce_{ctx16.layout_name}_kernel" if compact and fits else "error_reduce_kernel")
        finally:
            ctx16.set_reduce_compact(True)
            ctx16.set_tuning(0)
        assert_final_equal(got, want)
        outs.append(got)
    import torch

    for k in ("rate", "thr", "code", "germ_present"):
        assert torch.equal(getattr(outs[0], k).view(torch.uint8), getattr(outs[1], k).view(torch.uint8)), k


@pytest.mark.parametrize("slim", [False, True])
@pytest.mark.parametrize("P,S,n", [(4097, 64, 8), (1000, 37, 3), (20000, 256, 2)])
def test_sliced_merge_through_the_compact_kernel(ctx16, P, S, n, slim):
    """A shard of a multi-GPU panel on uint16 records goes through error_reduce_u16_kernel too (sums and germ-max pairs straight
    into the slice-major exchange buffers, both formats of the sums): the merged table is the single pass's, with the compact
    kernel and with the general one."""
    import torch

    from amplisolve_amd.dist import shard_range, slice_geometry, slice_planes

    recs = synth_recs(P, S)
    rng = np.random.default_rng(P + S)
    e = edge_case_recs(P, S, rng)
    pick = rng.random((S, P)) < 0.2
    recs[pick] = e[pick]
    ref = orc.error_finalize(orc.error_reduce(recs, P))
    L, _, _, bb = slice_geometry(P, n, slim)
    pl = slice_planes(slim)
    ctx16.set_slice_format(slim)
    ctx16.set_tuning(1, groups=1)  # one sample split, one lane group: the shape the compact kernel takes (uint16 records only)
    try:
        for compact in (True, False):
            ctx16.set_reduce_compact(compact)
            sums, gms = [], []
            for r in range(n):
                a, b = shard_range(S, r, n)
                s = torch.zeros(n * pl * L, dtype=torch.float64, device="cuda")
                g = torch.zeros(n * 8 * L, dtype=torch.float32, device="cuda")
                ctx16.error_reduce_sliced(to16(ctx16, recs[a:b]), P, n, s, g, first_sample=a)
                assert ctx16.last_reduce_kernel() == (f"error_reduce_{ctx16.layout_name}_kernel" if compact else "error_reduce_kernel")
                sums.append(s)
                gms.append(g)
            assert ctx16.flags() == 0
            total = torch.stack(sums).sum(0).view(n, pl * L)
            blocks = torch.zeros(n * bb, dtype=torch.uint8, device="cuda")
            for k in range(n):
                recv = torch.stack([g.view(n, 8 * L)[k] for g in gms]).contiguous()
                ctx16.error_finalize_slice(P, n, k, total[k].contiguous(), recv, blocks[k * bb:(k + 1) * bb])
            assert_final_equal(ctx16.error_table_unslice(P, n, blocks), ref)
    finally:
        ctx16.set_reduce_compact(True)
        ctx16.set_tuning(0)
        ctx16.set_slice_format(False)


@pytest.mark.parametrize("P,S,T,cut", [(5000, 24, 6, 2496), (20000, 256, 8, 9984), (130, 9, 3, 64)])
def test_position_ranges_on_two_streams_equal_the_single_pass(ctx16, P, S, T, cut):
    """The range split by hand (round 4's `two_ranges` block of bench.py, tools/split_probe.py): the panel's positions cut into two tile-aligned ranges, each with its own context, stream and
    outputs, reading VIEWS of the same resident arrays (row stride = the whole panel): error table and call mask of the two
    ranges side by side are the single pass's, bit for bit -- and both are the oracle's."""
    import torch

    from amplisolve_amd import Context

    lay = ctx16.layout_name
    recs, trecs = synth_recs(P, S), synth_recs(P, T, tumour=True)
    rng = np.random.default_rng(P + S + T)
    e = edge_case_recs(P, S, rng)
    pick = rng.random((S, P)) < 0.2
    recs[pick] = e[pick]
    ref = synth_ref(P)
    nd, td, rd = to16(ctx16, recs), to16(ctx16, trecs), _t(ref)
    relem = nd.shape[-1]
    want = orc.error_finalize(orc.error_reduce(recs, P, 0.002, 100))
    whole = ctx16.error_estimate(nd, P, 0.002, 100)
    assert_final_equal(whole, want)
    wres = ctx16.poisson_call(td, P, whole.thr, rd, 100, capacity=1 << 16)
    parts = []
    for lo, hi in ((0, cut), (cut, P)):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            c = Context(0)
            c.set_record_layout(lay)
            nrec = c.records(nd.view(S, P, relem)[:, lo:hi], lay, S, row_stride=P)
            trec = c.records(td.view(T, P, relem)[:, lo:hi], lay, T, row_stride=P)
            f = c.error_reduce_records(nrec, hi - lo, None, finalize=True)
            r = c.poisson_call_records(trec, hi - lo, f.thr, rd[lo:hi].contiguous(), 100, capacity=1 << 16)
        parts.append((c, f, r))
    torch.cuda.synchronize()
    for k in ("rate", "thr", "code", "germ_present", "germ_val"):
        got = torch.cat([getattr(f, k) for _, f, _ in parts], dim=-1)
        assert torch.equal(got.view(torch.uint8), getattr(whole, k).view(torch.uint8)), k
    assert torch.equal(torch.cat([r["call_mask"] for _, _, r in parts], dim=1), wres["call_mask"])
    assert sum(c.n_calls_total(r) for c, _, r in parts) == ctx16.n_calls_total(wres)
    for c, _, _ in parts:
        assert c.flags() == 0
        c.close()


@pytest.mark.parametrize("n", [2, 3, 4])
@pytest.mark.parametrize("P,S,T", [(5000, 24, 6), (20000, 64, 8), (100_000, 32, 12), (520, 9, 3)])
def test_position_ranges_inside_the_library(ctx16, P, S, T, n):
    """The shipped form of the range split (ampli_set_ranges, round 5): ampli_error_estimate and ampli_poisson_call cut the panel into
    n tile-aligned ranges on n streams inside the library -- the same output arrays, the call list's shards dealt to the ranges.
    Three passes back to back (the section stays open across them), then one joining call: error table, call mask and call list of
    the unsplit pass, bit for bit, which are the oracle's (both compact layouts: uint16 and 24-bit records)."""
    import torch

    lay = ctx16.layout_name
    recs, trecs = synth_recs(P, S), synth_recs(P, T, tumour=True)
    rng = np.random.default_rng(P + S + T + n)
    e = edge_case_recs(P, S, rng)
    pick = rng.random((S, P)) < 0.2
    recs[pick] = e[pick]
    ref = synth_ref(P)
    nd, td, rd = to16(ctx16, recs), to16(ctx16, trecs), _t(ref)
    want = orc.error_finalize(orc.error_reduce(recs, P, 0.002, 100))
    ctx16.set_tuning(1, groups=1)  # the compact kernel's shape also for the small panels of this test
    try:
        whole = ctx16.error_estimate(nd, P, 0.002, 100)
        assert_final_equal(whole, want)
        wres = ctx16.poisson_call(td, P, whole.thr, rd, 100, capacity=1 << 16)
        wcalls = ctx16.read_calls(wres)
        ctx16.set_ranges(n)
        assert ctx16.ranges_concurrent() in (True, False)  # whether every pair of the ranges' streams was seen to overlap: speed, not results
        fin = ctx16.error_estimate(nd, P, 0.002, 100)
        res = ctx16.poisson_call(td, P, fin.thr, rd, 100, capacity=1 << 16)
        ev = [[ctx16.event() for _ in range(3)] for _ in range(n)]
        for it in range(2):  # the same buffers again, no join in between: each range's stream keeps its own order
            for k in range(n):
                ctx16.range_record(k, ev[k][0])
            ctx16.error_estimate(nd, P, 0.002, 100, out=fin)
            for k in range(n):
                ctx16.range_record(k, ev[k][1])
            ctx16.poisson_call(td, P, fin.thr, rd, 100, call_mask=res["call_mask"], capacity=res["capacity"], calls_buf=res["calls_buf"], n_calls=res["n_calls"])
            for k in range(n):
                ctx16.range_record(k, ev[k][2])
        if (P + 63) // 64 >= 2 * n:  # every range's share of the calls took time on its own stream
            assert all(ctx16.elapsed_ms(ev[k][1], ev[k][2]) > 0 for k in range(n))
            assert all(ctx16.elapsed_ms(ev[k][0], ev[k][1]) > 0 for k in range(n))
        assert ctx16.flags() == 0  # joins
        for k in ("rate", "thr", "code", "germ_present", "germ_val"):
            assert torch.equal(getattr(fin, k).view(torch.uint8), getattr(whole, k).view(torch.uint8)), k
        assert torch.equal(res["call_mask"], wres["call_mask"])
        key = lambda c: (c["sample"], c["record"], c["alt"])
        got = ctx16.read_calls(res)
        assert sorted(map(key, got)) == sorted(map(key, wcalls)) and len(got) == ctx16.n_calls_total(res)
        assert [(c["q_fw"], c["q_bw"], c["af"]) for c in sorted(got, key=key)] == [(c["q_fw"], c["q_bw"], c["af"]) for c in sorted(wcalls, key=key)]
    finally:
        ctx16.set_ranges(1)
        ctx16.set_tuning(0)


def test_ranges_leave_other_shapes_alone(ctx16):
    """Launches outside the range shape run unsplit behind a join: positions listed more than once, a row length that is not a
    multiple of 4, the all-scores mode -- same results as a context without ranges."""
    import torch

    from amplisolve_amd.api import POISSON_FULL

    P, S, T, E = 4099, 12, 5, 0
    recs, trecs = synth_recs(P, S), synth_recs(P, T, tumour=True)
    ref = synth_ref(P)
    nd, td, rd = to16(ctx16, recs), to16(ctx16, trecs), _t(ref)
    ctx16.set_tuning(1, groups=1)
    try:
        whole = ctx16.error_estimate(nd, P, 0.002, 100)
        wres = ctx16.poisson_call(td, P, whole.thr, rd, 100, capacity=1 << 16)
        wfull = ctx16.poisson_call(td, P, whole.thr, rd, 100, mode=POISSON_FULL, capacity=1 << 16)
        ctx16.set_ranges(2)
        fin = ctx16.error_estimate(nd, P, 0.002, 100)  # split (uint16) or not (24-byte records)
        res = ctx16.poisson_call(td, P, fin.thr, rd, 100, capacity=1 << 16)  # P + E = 4099 is not a multiple of 4: unsplit, behind a join
        full = ctx16.poisson_call(td, P, fin.thr, rd, 100, mode=POISSON_FULL, capacity=1 << 16)
        assert ctx16.flags() == 0
        assert torch.equal(fin.thr.view(torch.int32), whole.thr.view(torch.int32))
        assert torch.equal(res["call_mask"], wres["call_mask"]) and torch.equal(full["call_mask"], wfull["call_mask"])
        assert ctx16.n_calls_total(res) == ctx16.n_calls_total(wres)
        ctx16.set_ranges(1)  # without ranges, "range 0" is the context's own stream
        ea, eb = ctx16.event(), ctx16.event()
        ctx16.range_record(0, ea)
        ctx16.error_estimate(nd, P, 0.002, 100, out=fin)
        ctx16.range_record(0, eb)
        assert ctx16.elapsed_ms(ea, eb) > 0
    finally:
        ctx16.set_ranges(1)
        ctx16.set_tuning(0)
