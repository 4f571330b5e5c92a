"""Error behaviour at the C ABI (include/amplisolve_hip.h): a bad argument is answered with a negative status and a message
(ampli_last_error) BEFORE anything is launched -- never with a kernel that reads where it should not -- and the context stays
usable: the same context then runs a valid pass and matches the oracle.  The reference's own error behaviour lives one level up,
in the command lines (messages + exit status 0, tests/test_gpu_cli.py); this file is about the entry points a binding would call."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyoracle as orc
from tests.helpers import synth_recs
from tests.test_gpu_parity import _t, assert_final_equal
from tests.test_gpu_records import _pack

pytestmark = pytest.mark.gpu
E_INVALID = -1
P, S = 320, 12


def _refused(ctx, rc, code=E_INVALID):
    assert rc == code, (rc, ctx.lib.ampli_last_error(ctx.h).decode())
    assert ctx.lib.ampli_last_error(ctx.h).decode() != ""


def _valid_pass_matches_the_oracle(ctx, recs, lay="u16"):
    ctx.set_record_layout(lay)
    try:
        got = ctx.error_estimate(_pack(ctx, recs, lay), P, 0.002, 100)
        assert ctx.flags() == 0
        assert_final_equal(got, orc.error_finalize(orc.error_reduce(recs, P, 0.002, 100)))
    finally:
        ctx.set_record_layout("i32")


@pytest.fixture()
def cohort(ctx):
    import torch

    recs = np.minimum(synth_recs(P, S), 65534).astype(np.int32)
    ctx.set_record_layout("u16")
    packed = _pack(ctx, recs, "u16")
    ctx.set_record_layout("i32")
    out = ctx._new_error_table(P)
    yield recs, packed, out, torch
    ctx.set_ranges(1)
    ctx.set_reduce_compact(True)


def _reduce(ctx, rec, out, acc=None, how=0, P_=P, cov=100):
    return ctx.lib.ampli_error_reduce_records(ctx.h, C.byref(rec), P_, 0, 0.002, cov, C.byref(acc.struct) if acc is not None else None, how,
                                              out.rate.data_ptr(), out.code.data_ptr(), out.thr.data_ptr(), out.germ_val.data_ptr(),
                                              out.germ_present.data_ptr(), out.flags.data_ptr())


def test_error_reduce_records_refuses_what_it_cannot_run(ctx, cohort):
    recs, packed, out, torch = cohort
    good = ctx.records(packed, "u16", S)
    assert _reduce(ctx, good, out) == 0
    # the cohort description
    for change in (dict(n_samples=0), dict(n_samples=-3), dict(E=-1), dict(layout=7), dict(recs=None), dict(row_stride=P - 1),
                   dict(E=5),  # extras without dup_off (and a row stride that cannot hold them)
                   dict(recs=packed.data_ptr() + 4)):  # not 16-byte aligned
        r = ctx.records(packed, "u16", S)
        for k, v in change.items():
            setattr(r, k, v)
        _refused(ctx, _reduce(ctx, r, out))
    r = ctx.records(packed, "u16", S, E=4, row_stride=P + 4)  # E > 0 but no dup_off
    _refused(ctx, _reduce(ctx, r, out))
    # the call
    _refused(ctx, _reduce(ctx, good, out, P_=0))
    _refused(ctx, _reduce(ctx, good, out, P_=-5))
    _refused(ctx, _reduce(ctx, good, out, cov=0))
    _refused(ctx, _reduce(ctx, good, out, how=1))  # accumulate without a table
    _refused(ctx, _reduce(ctx, good, out, how=3))
    other = ctx.new_acc(P + 64)  # a table bound for another panel
    _refused(ctx, _reduce(ctx, good, out, acc=other))
    unbound = ctx.new_acc(P)
    unbound.struct.snt = None
    _refused(ctx, _reduce(ctx, good, out, acc=unbound))
    # rate without code; neither outputs nor table
    rc = ctx.lib.ampli_error_reduce_records(ctx.h, C.byref(good), P, 0, 0.002, 100, None, 0, out.rate.data_ptr(), None, None, None, None, None)
    _refused(ctx, rc)
    rc = ctx.lib.ampli_error_reduce_records(ctx.h, C.byref(good), P, 0, 0.002, 100, None, 0, None, None, None, None, None, None)
    _refused(ctx, rc)
    assert ctx.lib.ampli_error_reduce_records(None, C.byref(good), P, 0, 0.002, 100, None, 0, None, None, None, None, None, None) == E_INVALID
    ctx.sync()
    assert ctx.flags() == 0
    _valid_pass_matches_the_oracle(ctx, recs)


def test_sliced_store_and_ranges_refuse_bad_arguments(ctx, cohort):
    recs, packed, out, torch = cohort
    good = ctx.records(packed, "u16", S)
    ctx.set_slice_group(1, 0)
    ctx.set_slice_format(False)  # 21 planes of sums: the larger of the two formats
    L = ctx.slice_len(P, 2)
    sums = torch.zeros(2 * 21 * L, dtype=torch.float64, device=ctx.device)
    gm = torch.zeros(2 * 8 * L, dtype=torch.float32, device=ctx.device)

    def sliced(n=2, s=sums.data_ptr(), g=gm.data_ptr(), how=2, acc=None):
        return ctx.lib.ampli_error_reduce_records_sliced(ctx.h, C.byref(good), P, 0, 0.002, 100, C.byref(acc.struct) if acc is not None else None, how, n, s, g)

    assert sliced() == 0
    _refused(ctx, sliced(n=0))
    _refused(ctx, sliced(s=None))
    _refused(ctx, sliced(g=None))
    _refused(ctx, sliced(how=3))  # accumulate without a table
    _refused(ctx, sliced(acc=ctx.new_acc(P + 64)))
    # position ranges: 1 .. 4; an event needs a range that exists
    for n in (0, -1, 5, 100):
        _refused(ctx, ctx.lib.ampli_set_ranges(ctx.h, n))
    ev = ctx.event()
    assert ctx.lib.ampli_range_event_record(ctx.h, 1, ev) == E_INVALID  # no ranges set: only range 0 exists
    assert ctx.lib.ampli_range_event_record(ctx.h, -1, ev) == E_INVALID
    assert ctx.lib.ampli_range_event_record(ctx.h, 0, None) == E_INVALID
    assert ctx.lib.ampli_range_event_record(ctx.h, 0, ev) == 0
    ctx.set_ranges(2)
    assert ctx.lib.ampli_range_event_record(ctx.h, 2, ev) == E_INVALID
    assert ctx.lib.ampli_range_event_record(ctx.h, 1, ev) == 0
    assert ctx.lib.ampli_set_ranges(None, 2) == E_INVALID and ctx.lib.ampli_ranges_join(None) == E_INVALID
    ctx.set_ranges(1)
    ctx.sync()
    assert ctx.flags() == 0
    _valid_pass_matches_the_oracle(ctx, recs)


def test_poisson_call_refuses_what_it_cannot_run(ctx, cohort):
    recs, packed, out, torch = cohort
    T = 5
    trecs = np.minimum(synth_recs(P, T), 65534).astype(np.int32)
    ctx.set_record_layout("u16")
    tp = _pack(ctx, trecs, "u16")
    ctx.set_record_layout("i32")
    table = orc.error_finalize(orc.error_reduce(recs, P, 0.002, 100))
    thr, ref = _t(np.ascontiguousarray(table["thr"], dtype=np.float32)), _t(np.zeros(P, np.uint8))
    mask = torch.zeros(T * P, dtype=torch.uint8, device=ctx.device)
    calls = torch.zeros(64 * 64, dtype=torch.uint8, device=ctx.device)
    n_calls = torch.zeros(64, dtype=torch.int64, device=ctx.device)
    q = torch.zeros(T * P * 8, dtype=torch.float64, device=ctx.device)
    good = ctx.records(tp, "u16", T)

    def call(rec=good, P_=P, thr_=thr.data_ptr(), ref_=ref.data_ptr(), cov=100, mode=1, mask_=mask.data_ptr(), calls_=None, cap=0, n_=None, q_=None):
        return ctx.lib.ampli_poisson_call_records(ctx.h, C.byref(rec), P_, thr_, ref_, cov, mode, mask_, calls_, cap, n_, q_, None)

    prefilter = 1
    assert call(mode=prefilter) == 0
    _refused(ctx, call(P_=0))
    _refused(ctx, call(thr_=None))
    _refused(ctx, call(ref_=None))
    _refused(ctx, call(mask_=None))
    _refused(ctx, call(cov=0))
    _refused(ctx, call(mode=9))
    _refused(ctx, call(mode=prefilter, q_=q.data_ptr()))  # dense scores need the all-scores mode
    _refused(ctx, call(calls_=calls.data_ptr(), cap=64, n_=None))  # a call list needs its counters
    _refused(ctx, call(calls_=calls.data_ptr(), cap=3, n_=n_calls.data_ptr()))  # ... and at least one slot per shard
    for change in (dict(n_samples=0), dict(layout=5), dict(recs=None), dict(recs=tp.data_ptr() + 8), dict(E=2)):
        r = ctx.records(tp, "u16", T)
        for k, v in change.items():
            setattr(r, k, v)
        _refused(ctx, call(rec=r))
    assert call(P_=1 << 30) < 0  # beyond the kernels' index range (AMPLI_E_RANGE), refused before any launch
    ctx.sync()
    assert ctx.flags() == 0
    _valid_pass_matches_the_oracle(ctx, recs)
