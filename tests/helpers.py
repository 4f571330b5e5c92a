"""Shared test helpers (host-side synthetic panels through the product's host library)."""
import ctypes as C

import numpy as np

from amplisolve_amd import host_lib

SEED = 0xA3F15017


def synth_recs(P, n, first=0, seed=SEED, depth=2000, tumour=False):
    out = np.empty((n, P, 8), np.int32)
    rc = host_lib().ampli_host_synth_fill(out.ctypes.data_as(C.c_void_p), P, n, first, seed, depth, int(tumour))
    assert rc == 0
    return out


def synth_ref(P, seed=SEED):
    out = np.empty((P,), np.uint8)
    assert host_lib().ampli_host_synth_ref(out.ctypes.data_as(C.c_void_p), P, seed) == 0
    return out


def edge_case_recs(P, S, rng):
    """Random records that hit every branch of the gate: absent cells, FW or BW == 0, depth around the
    coverage cutoff, AF around 5 %, homozygous alts, zero counts."""
    recs = np.zeros((S, P, 8), np.int32)
    depth = rng.choice([0, 1, 50, 99, 100, 101, 200, 1000, 5000, 33395], size=(S, P, 2))
    for st in range(2):
        d = depth[:, :, st]
        frac = rng.choice([0.0, 0.0005, 0.002, 0.01, 0.0499, 0.05, 0.0501, 0.06, 0.3, 0.5, 1.0], size=(S, P, 3))
        alts = np.minimum((d[:, :, None] * frac).astype(np.int64) + rng.integers(0, 2, size=(S, P, 3)), d[:, :, None])
        # make the alts fit: scale down when they exceed the depth
        tot = alts.sum(-1)
        over = tot > d
        alts[over] = 0
        ref = d - alts.sum(-1)
        ref_nt = rng.integers(0, 4, size=(P,))
        for p in range(P):
            order = [nt for nt in range(4) if nt != ref_nt[p]]
            recs[:, p, st * 4 + ref_nt[p]] = ref[:, p]
            for j, nt in enumerate(order):
                recs[:, p, st * 4 + nt] = alts[:, p, j]
    absent = rng.random((S, P)) < 0.1
    recs[absent] = 0
    recs[absent, 0] = np.iinfo(np.int32).min
    return recs


def borderline_triples(want=12, seed=5, gate=5.0, eps=1e-6, errs=(0.000001, 0.000002, 0.000003, 0.000005, 0.000007, 0.000011, 0.000013)):
    """(k, RD, err, Q) with the REFERENCE's Q (the oracle's scorer is bit-identical to it, tests/test_oracle_golden.py) within
    eps of the gate, on both sides of it.  err has at most six decimals so that it survives the error table's "%f"; the fine
    knob is RD: Q falls as RD grows, bisection finds the crossing and its neighbours are kept when they are close enough."""
    from oracle import pyoracle as orc

    L = orc.lib()
    rng = np.random.default_rng(seed)
    out = []
    tries = 0
    while len(out) < want and tries < 4000:
        tries += 1
        err = float(np.float32(rng.choice(errs)))
        k = int(rng.integers(3, 60))
        lo, hi = 100, (1 << 31) - 2  # Q(lo) high (tiny mean), Q(hi) low
        if not (float(L.oracle_score(k, lo, err)) >= gate > float(L.oracle_score(k, hi, err))):
            continue
        while hi - lo > 1:
            mid = (lo + hi) // 2
            if float(L.oracle_score(k, mid, err)) >= gate:
                lo = mid
            else:
                hi = mid
        for rd in (lo - 1, lo, hi, hi + 1):
            q = float(L.oracle_score(k, rd, err))
            if abs(q - gate) <= eps and (k, rd, err) not in [(a, b, c) for a, b, c, _ in out]:
                out.append((k, rd, err, q))
    return out
