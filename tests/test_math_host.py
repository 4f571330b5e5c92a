"""Scalar helpers the kernels share with the host (csrc/ampli_math.h), checked on the CPU against the
literal reference formulations in the oracle.  These pin the two transformations the kernels rely on:
the integer AF gate and the integer text round trip; and the exact-decision bound of the prefilter."""
import ctypes as C

import numpy as np

from amplisolve_amd import host_lib
from oracle import pyoracle as orc


def test_af_gate_integer_bound_equals_fp_gate():
    """x <= floor(d * 26843545 / 2^29)  <=>  (double)((float)x/(float)d) <= 0.05   for 0 < d < 2^24."""
    H, O = host_lib(), orc.lib()
    rng = np.random.default_rng(0)
    ds = np.unique(np.concatenate([np.arange(1, 6000), rng.integers(1, 1 << 24, 30000), [(1 << 24) - 1, 1 << 23, 12345678]]))
    for d in ds:
        d = int(d)
        lim = H.ampli_host_af_limit(d)
        # the bound is tight: lim passes, lim+1 fails; and it is monotone so nothing else needs checking
        assert O.oracle_af_gate(lim, d) == 1, d
        assert O.oracle_af_gate(lim + 1, d) == 0, d
        assert abs(lim - 0.05 * d) <= 1
    for d in (100, 101, 1999, 33395):  # exhaustive in x for a few depths
        lim = H.ampli_host_af_limit(d)
        for x in range(0, d + 1):
            assert O.oracle_af_gate(x, d) == (1 if x <= lim else 0)


def test_text_roundtrip_matches_sprintf_strtof():
    H, O = host_lib(), orc.lib()
    rng = np.random.default_rng(1)
    vals = np.concatenate([
        rng.random(300000, dtype=np.float32) * np.float32(0.07),
        rng.random(50000, dtype=np.float32) * np.float32(20),
        (np.float32(10.0) ** rng.uniform(-14, 4, 50000)).astype(np.float32),
        (np.arange(0, 70000, dtype=np.float64) * 1e-6 + 5e-7).astype(np.float32),        # decimal ties
        (np.arange(0, 70000, dtype=np.float64) * 1e-6).astype(np.float32),                # exact 6-decimal values
        np.array([0, -0.0, 1e-7, 4.9e-7, 5e-7, 5.1e-7, 1.5e-6, 0.002, 0.01, 0.05, 1, 15.999999, 16, 17.5, 1e6, 1e-45, 3e38, -0.0021], np.float32),
    ]).astype(np.float32)
    out = np.empty_like(vals)
    H.ampli_host_text_roundtrip_batch(vals.ctypes.data_as(C.c_void_p), vals.size, out.ctypes.data_as(C.c_void_p))
    exp = np.array([O.oracle_text_roundtrip(float(v)) for v in vals], np.float32)
    assert np.array_equal(out.view(np.int32), exp.view(np.int32))
    # every float in a dense window around typical rates (C = 0.002 .. 0.01)
    lo = np.float32(0.0019).view(np.int32)
    win = np.arange(lo, lo + 200000, dtype=np.int32).view(np.float32)
    out = np.empty_like(win)
    H.ampli_host_text_roundtrip_batch(win.ctypes.data_as(C.c_void_p), win.size, out.ctypes.data_as(C.c_void_p))
    exp = np.array([O.oracle_text_roundtrip(float(v)) for v in win], np.float32)
    assert np.array_equal(out.view(np.int32), exp.view(np.int32))


def test_prefilter_bound_is_exact_decision():
    """AMPLI_POISSON_PREFILTER skips a score when k <= m = RD*err.  That is only legal if the REFERENCE's scorer
    (with its 99-step caps) returns Q < 5 there, so VC:898 is false anyway.  Sweep k <= m over five decades."""
    H = host_lib()
    rng = np.random.default_rng(4)
    errs = np.array([0.0001, 0.0005, 0.001, 0.002, 0.002189, 0.0035, 0.01, 0.02, 0.05, 0.25, 0.0], np.float32)
    ks, rds, es = [], [], []
    for err in errs:
        e = float(err) if err != 0 else float(np.float32(0.0010008))
        rd = np.unique(np.concatenate([np.arange(1, 3000, 7), rng.integers(1, 1 << 24, 12000), np.geomspace(10, (1 << 24) - 1, 2000).astype(np.int64)]))
        m = rd * e
        for frac in (1.0, 0.999, 0.99, 0.9, 0.75, 0.5, 0.25, 0.1, 0.01):
            for j in (0, 1, 2, 5):
                k = np.floor(m * frac).astype(np.int64) - j
                sel = k >= 1
                ks.append(k[sel]); rds.append(rd[sel]); es.append(np.full(sel.sum(), err, np.float32))
    k = np.concatenate(ks).astype(np.int32); rd = np.concatenate(rds).astype(np.int32); er = np.concatenate(es)
    assert k.size > 500000
    skip = np.array([H.ampli_host_prefilter_nocall(int(a), int(b), float(c)) for a, b, c in zip(k[::37], rd[::37], er[::37])])
    assert skip.all()  # all of these have k <= m: the product would skip them
    q, _ = orc.score_batch(k, rd, er)
    assert not np.isnan(q).any()
    assert q.max() < 5.0, (q.max(), k[q.argmax()], rd[q.argmax()], er[q.argmax()])
    assert q.max() < 3.1  # P(X >= k) > 0.49 for k <= mean: Q stays near 3
    # and the complement is not skipped
    assert H.ampli_host_prefilter_nocall(3, 1000, 0.002) == 0 and H.ampli_host_prefilter_nocall(2, 1000, 0.002) == 1
    assert H.ampli_host_prefilter_nocall(5, 1000, -1.0) == 1 and H.ampli_host_prefilter_nocall(0, 1000, 0.002) == 1


def test_fp32_prefilter_is_conservative():
    """The streaming kernel's fp32 test may only skip what the exact bound skips (k <= RD*err in double)."""
    H = host_lib()
    rng = np.random.default_rng(8)
    errs = np.array([0.0001, 0.0005, 0.001, 0.002, 0.002189, 0.0035, 0.01, 0.02, 0.05, 0.25, 0.0, -1.0, 0.7, 3e-6], np.float32)
    n_skip = n_keep = 0
    for err in errs:
        e = float(np.float32(0.0010008)) if err == 0 else float(err)
        rd = np.unique(np.concatenate([rng.integers(0, 1 << 24, 3000), np.arange(0, 400), [(1 << 24) - 1, 1 << 24, 1 << 25]]))
        for d in rd:
            m = d * e
            for k in {0, 1, int(m) - 1, int(m), int(m) + 1, int(m) + 2, int(m * 0.999999), int(m * 1.000001) + 1, (1 << 24) - 1, 1 << 24}:
                if k < 0:
                    continue
                sk = H.ampli_host_prefilter_skip_f32(int(k), int(d), float(err))
                if sk:
                    n_skip += 1
                    assert H.ampli_host_prefilter_nocall(int(k), int(d), float(err)) == 1, (k, d, err)
                else:
                    n_keep += 1
    assert n_skip > 50000 and n_keep > 50000
    # and it is not vacuous: a count well below the mean is skipped, one above is kept
    assert H.ampli_host_prefilter_skip_f32(1, 1000, 0.002) == 1 and H.ampli_host_prefilter_skip_f32(3, 1000, 0.002) == 0


def test_af_gate_integer_bound_exhaustive_over_all_depths():
    """EVERY depth the integer gate is used for (1 <= d < 2^24): the bound is tight -- x = limit passes the
    reference's fp32 test, x = limit + 1 fails it; the test is monotone in x, so this is the whole equivalence."""
    H, O = host_lib(), orc.lib()
    d = np.arange(1, 1 << 24, dtype=np.int32)
    lim = np.empty_like(d)
    H.ampli_host_af_limit_batch(d.ctypes.data_as(C.c_void_p), d.size, lim.ctypes.data_as(C.c_void_p))
    out = np.empty(d.size, np.uint8)
    O.oracle_af_gate_batch(lim.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), C.c_int64(d.size), out.ctypes.data_as(C.c_void_p))
    assert out.all()
    lim1 = lim + 1
    O.oracle_af_gate_batch(lim1.ctypes.data_as(C.c_void_p), d.ctypes.data_as(C.c_void_p), C.c_int64(d.size), out.ctypes.data_as(C.c_void_p))
    assert not out.any()
    assert (np.abs(lim - 0.05 * d.astype(np.float64)) <= 1).all()


def test_af_limit_as_one_float_multiply_is_the_integer_limit_for_every_depth():
    """error_reduce_u16_kernel forms the AF limit as (int)(float(d) * 0x1.999998p-5f) (full rate; the float of d is needed anyway):
    the same integer as floor(d * 26843545 / 2^29) for EVERY depth the fast kernels accept, 0 <= d < 2^24."""
    H = host_lib()
    d = np.arange(0, 1 << 24, dtype=np.int32)
    a, b = np.empty_like(d), np.empty_like(d)
    H.ampli_host_af_limit_batch(d.ctypes.data_as(C.c_void_p), d.size, a.ctypes.data_as(C.c_void_p))
    H.ampli_host_af_limit_f32_batch(d.ctypes.data_as(C.c_void_p), d.size, b.ctypes.data_as(C.c_void_p))
    assert np.array_equal(a, b)
    assert np.array_equal(a.astype(np.uint64), (d.astype(np.uint64) * 26843545) >> 29)
    # an absent record's strand sum is negative: whatever the limit is then, it is not positive (callers mask it anyway)
    neg = np.array([-1, -100, -(1 << 23), -(2**31) + 5], np.int32)
    out = np.empty_like(neg)
    H.ampli_host_af_limit_f32_batch(neg.ctypes.data_as(C.c_void_p), neg.size, out.ctypes.data_as(C.c_void_p))
    assert (out <= 0).all()


def test_division_free_series_of_the_drain_kernel_is_the_reference_series():
    """The drain kernel scores queued items (k > m) with the series of VC:3785-3794 rewritten without divisions and
    without the early exit -- since round 6 with kf_lgamma(k + 1) from the table of its own values and the prefactor multiplied
    by the sum instead of adding its logarithm (ampli_drain_p: a third fewer instructions on the drain's critical path).
    Against the oracle's literal scorer: p within 1e-11 relative everywhere (the contract is 1e-6), the Q = 100 / Q >= 5 /
    Q >= 20 decisions identical outside 1e-9 of a gate (a pair within 1e-6 of the call gate is re-decided by the host either
    way) -- including z so close to s that the reference's 99-term cap truncates the series, and depths up to the int32 range."""
    H = host_lib()
    rng = np.random.default_rng(8)
    ks, rds, es = [], [], []
    for err in (0.0001, 0.0005, 0.002, 0.002189, 0.0035, 0.01, 0.02, 0.05, 0.25):
        for scale in (50, 300, 1000, 2500, 25000, 400000, 3_000_000, 16_000_000, 500_000_000):
            rd = rng.integers(max(1, scale // 2), scale * 2, 600)
            m = rd * float(np.float32(err))
            # k from just above m (slow convergence, cap region) to far above it (Q = 100)
            for f in (1.0, 1.01, 1.1, 1.5, 3.0, 10.0):
                k = np.maximum(np.floor(m * f).astype(np.int64) + rng.integers(1, 4, rd.size), 1)
                ok = k < (1 << 31) - 1
                ks.append(k[ok]); rds.append(rd[ok]); es.append(np.full(ok.sum(), err))
    k = np.concatenate(ks).astype(np.int32)
    rd = np.concatenate(rds).astype(np.int32)
    e = np.concatenate(es).astype(np.float32)
    keep = k.astype(np.float64) > rd.astype(np.float64) * e.astype(np.float64)
    k, rd, e = k[keep], rd[keep], e[keep]
    q = np.empty(k.size, np.float64)
    p = np.empty(k.size, np.float64)
    H.ampli_host_drain_score_batch(k.ctypes.data_as(C.c_void_p), rd.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p), k.size,
                                   q.ctypes.data_as(C.c_void_p), p.ctypes.data_as(C.c_void_p))
    qo, po = orc.score_batch(k, rd, e)
    assert k.size > 200_000
    # p = 1 - (1 - P) is a multiple of 2^-53: a last-bit difference in P can move it by one such step
    excess = np.abs(p - po) - (1e-11 * po + 2.3e-16)
    assert np.max(excess) <= 0, (np.argmax(excess), np.max(excess))
    for thr in (5.0, 20.0, 100.0):
        away = np.abs(qo - thr) > 1e-9
        assert np.array_equal((q >= thr)[away], (qo >= thr)[away]), thr
    fin = qo < 100
    assert np.max(np.abs(q[fin] - qo[fin])) < 1e-7


def test_dense_scorer_of_the_all_scores_mode_is_the_reference_scorer():
    """AMPLI_POISSON_FULL scores every (record, alt, strand) with the integer-count form of the reference's recipe (round 4:
    lgamma at the integers, division-free series, continued fraction through its convergents; csrc/ampli_math.h).  Against the
    oracle's literal scorer -- itself bit-identical to the reference's compiled functions -- on the reference's own golden grid,
    on counts around the mean at every depth (both branches of kf_gammaq, the identity step j = s of the continued fraction,
    the 99-step caps of both loops, k and m far beyond 100) and on the special codes: p within 1e-11 relative + one step of
    1 - (1 - P) (the contract is 1e-6), the specials identical, the Q >= 5 / >= 20 / = 100 decisions identical outside 1e-9 of a gate."""
    import os

    H = host_lib()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vc_scorer_reference.npz"))
    rng = np.random.default_rng(21)
    ks, rds, es = [g["k"].astype(np.int64)], [g["rd"].astype(np.int64)], [g["err"].astype(np.float64)]
    for err in (0.0, -1.0, 0.0001, 0.0005, 0.002, 0.002189, 0.0035, 0.01, 0.02, 0.05, 0.25, 0.9):
        for scale in (1, 5, 50, 300, 1000, 2500, 25000, 400000, 3_000_000, 16_000_000, 500_000_000):
            rd = rng.integers(max(1, scale // 2), scale * 2 + 1, 400)
            m = rd * float(np.float32(err if err > 0 else 0.0010008))
            for f in (0.0, 0.3, 0.8, 0.97, 1.0, 1.03, 1.3, 3.0, 10.0):
                k = np.maximum(np.floor(m * f).astype(np.int64) + rng.integers(-2, 3, rd.size), 0)
                ok = k < (1 << 31) - 1
                ks.append(k[ok]); rds.append(rd[ok]); es.append(np.full(int(ok.sum()), err))
    # small counts x small means: what a 2000x panel holds
    kk, mm = np.meshgrid(np.arange(0, 40), np.arange(1, 120), indexing="ij")
    ks.append(kk.ravel()); rds.append(mm.ravel() * 50); es.append(np.full(kk.size, 0.002))
    k = np.concatenate(ks).astype(np.int32)
    rd = np.concatenate(rds).astype(np.int32)
    e = np.concatenate(es).astype(np.float32)
    q = np.empty(k.size, np.float64)
    H.ampli_host_dense_score_batch(k.ctypes.data_as(C.c_void_p), rd.ctypes.data_as(C.c_void_p), e.ctypes.data_as(C.c_void_p), k.size, q.ctypes.data_as(C.c_void_p))
    qo, po = orc.score_batch(k, rd, e)
    assert k.size > 400_000
    special = (qo == -888) | (qo == 100) | (qo == 0) | np.isnan(qo)
    assert np.array_equal(q[qo == -888], qo[qo == -888]) and np.array_equal(np.isnan(q), np.isnan(qo))
    for thr in (5.0, 20.0, 100.0):
        far = np.abs(qo - thr) > 1e-9
        assert np.array_equal((q >= thr)[far & ~np.isnan(qo)], (qo >= thr)[far & ~np.isnan(qo)]), thr
    assert np.array_equal(q[qo == 0] == 0, np.ones(int((qo == 0).sum()), bool))
    fin = ~special & (q != 100) & (q != 0)
    # Q = -10 log10 p: |dQ| = 4.34 |dp| / p
    pq = 10.0 ** (-q[fin] / 10)
    excess = np.abs(pq - po[fin]) - (1e-11 * po[fin] + 2.3e-16)
    assert np.max(excess) <= 0, (k[fin][np.argmax(excess)], rd[fin][np.argmax(excess)], e[fin][np.argmax(excess)], np.max(excess))
    assert np.max(np.abs(q[fin] - qo[fin])) < 1e-7


def test_decision_guard_is_the_reference_operation_sequence():
    """The host re-evaluates calls within 1e-6 of a gate with the reference's own sequence (double kf_gammaq, long double
    log10; csrc/host/annotate.cpp).  On the golden grid produced by the reference's compiled scorer it must return the very
    doubles, and its >= 5 / < 20 decisions must be the long-double ones."""
    import os

    H = host_lib()
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "vc_scorer_reference.npz"))
    k, rd, err, q = g["k"], g["rd"], g["err"], g["q"]
    ge5, lt20 = C.c_int32(), C.c_int32()
    idx = np.random.default_rng(3).choice(k.size, min(k.size, 20000), replace=False)
    for i in idx:
        got = H.ampli_host_guard_score(int(k[i]), int(rd[i]), float(err[i]), C.byref(ge5), C.byref(lt20))
        assert (got == q[i]) or (np.isnan(got) and np.isnan(q[i])), (k[i], rd[i], err[i], got, q[i])
        if not np.isnan(q[i]) and abs(q[i] - 5) > 1e-9 and abs(q[i] - 20) > 1e-9:
            assert ge5.value == int(q[i] >= 5) and lt20.value == int(q[i] < 20)
