"""poisson_call, one kernel (round 6) against stream + drain (rounds 1-5), on one box in one process, interleaved:
  * config 3 (100 k x 96 tumours, uint16 records) inside the alternating loop with error_estimate (the way bench.py's step sees it);
  * config 4's tumour shard: the 1024-tumour call against T / N = 512 / 256 / 128 tumours (what one of N GPUs runs) -> efficiency
    t(1024) / N / t(1024 / N), every call cold (the normals' reduce in between pushes the tumours out of the Infinity Cache).
PF_ROWS: rows per wave (0 = auto).  Usage: python tools/poisson_fused_ab.py [c3] [c4]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from amplisolve_amd import Context

SEED = 0xA3F15019
what = sys.argv[1:] or ["c3", "c4"]
torch.cuda.set_stream(torch.cuda.Stream())
ctx = Context(0)
ctx.set_record_layout("u16")
rows = int(os.environ.get("PF_ROWS", "0"))


def packed(P, n, tumour):
    out = []
    for lo in range(0, n, 128):  # int32 staging of 128 samples at a time
        raw = ctx.synth_fill(P, min(128, n - lo), seed=SEED, depth=2000, tumour=tumour, first_sample=lo)
        t, fits = ctx.pack(raw, "u16")
        assert fits
        out.append(t)
    return torch.cat(out) if len(out) > 1 else out[0]


def timed_call(t, P, fin, refc, res, normals, reps, fused):
    if hasattr(ctx, "set_poisson_fused"):  # only with tools/experiments/r06_poisson_fused_per_workgroup.patch applied
        ctx.set_poisson_fused(fused)
    elif fused:
        return float("nan"), float("nan"), float("nan")
    ctx.set_poisson_tuning(rows, 0)
    evs = [[ctx.event() for _ in range(2)] for _ in range(reps)]
    for i in range(reps + 2):
        ctx.error_estimate(normals, P, 0.002, 100, out=fin)  # cold tumours for the call
        if i >= 2:
            ctx.record(evs[i - 2][0])
        ctx.poisson_call(t, P, fin.thr, refc, 100, call_mask=res["call_mask"][: t.shape[0]], capacity=res["capacity"], calls_buf=res["calls_buf"], n_calls=res["n_calls"])
        if i >= 2:
            ctx.record(evs[i - 2][1])
    torch.cuda.synchronize()
    v = sorted(ctx.elapsed_ms(a, b) for a, b in evs)
    return v[len(v) // 2], v[0], sum(v) / len(v)


if "c3" in what:
    P, S, T = 100_000, 256, 96
    normals, tum, refc = packed(P, S, False), packed(P, T, True), ctx.synth_ref(P, seed=SEED)
    fin = ctx.error_estimate(normals, P, 0.002, 100)
    res = ctx.poisson_call(tum, P, fin.thr, refc, 100, capacity=1 << 20)
    n_ref = ctx.n_calls_total(res)
    call_bytes = 16 * P * T + 33 * P + P * T
    for rnd in range(3):
        for fused in (True, False):
            med, mn, mean = timed_call(tum, P, fin, refc, res, normals, 40, fused)
            assert ctx.n_calls_total(res) == n_ref and ctx.flags() == 0
            print(f"c3 round {rnd} {'one kernel ' if fused else 'two kernels'}: poisson_call median {med * 1e3:6.1f} us  min {mn * 1e3:6.1f}  mean {mean * 1e3:6.1f}  "
                  f"= {call_bytes / med / 1e9:5.2f} TB/s = {call_bytes / med / 1e9 / 8:5.3f} of peak  ({n_ref} calls)", flush=True)
    del normals, tum, res

if "c4" in what:
    P, S, T = 100_000, 256, 1024
    normals, tum, refc = packed(P, S, False), packed(P, T, True), ctx.synth_ref(P, seed=SEED)
    fin = ctx.error_estimate(normals, P, 0.002, 100)
    res = ctx.poisson_call(tum, P, fin.thr, refc, 100, capacity=1 << 22)
    for fused in (True, False):
        base = None
        for n in (1, 2, 4, 8):
            med, mn, mean = timed_call(tum[: T // n], P, fin, refc, res, normals, 20, fused)
            if n == 1:
                base = med
            print(f"c4 {'one kernel ' if fused else 'two kernels'}: N = {n}: {T // n:4d} tumours  poisson_call median {med * 1e3:7.1f} us (min {mn * 1e3:7.1f})   "
                  f"efficiency t(1024) / N / t = {base / n / med:5.3f}", flush=True)
ctx.close()
