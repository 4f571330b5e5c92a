"""What would folding config 3's thin second round of workgroups into the first buy?  error_estimate over 81 920 positions x 320 samples
(one full round of 1280 workgroups, 80 rows per wave: the work a folded launch would do per workgroup) against config 3's own 100 000 x
256 (1.22 rounds of 64 rows per wave) -- the same number of records; uint16 records, launches back to back and alternating with a
poisson_call of config 3's size (the bench loop's conditions)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

from amplisolve_amd import Context

SEED = 0xA3F15019
ctx = Context(0)
ctx.set_record_layout("u16")


def cohort(P, S):
    n = ctx.synth_fill(P, S, seed=SEED, depth=2000)
    return ctx.pack(n, "u16")[0]


PT, T = 100_000, 96
tum = ctx.pack(ctx.synth_fill(PT, T, seed=SEED, depth=2000, tumour=True), "u16")[0]
ref = ctx.synth_ref(PT, seed=SEED)
fin_t = ctx.error_estimate(cohort(PT, 32), PT, 0.002, 100)
res = ctx.poisson_call(tum, PT, fin_t.thr, ref, 100, capacity=1 << 20)
for P, S in ((100_000, 256), (81_920, 320), (81_920, 256), (102_400, 256)):
    n = cohort(P, S)
    fin = ctx.error_estimate(n, P, 0.002, 100)
    for alt in (False, True):
        reps = 30
        ev = [[ctx.event(), ctx.event()] for _ in range(reps)]
        for i in range(reps + 3):
            j = i - 3
            if j >= 0:
                ctx.record(ev[j][0])
            ctx.error_estimate(n, P, 0.002, 100, out=fin)
            if j >= 0:
                ctx.record(ev[j][1])
            if alt:
                ctx.poisson_call(tum, PT, fin_t.thr, ref, 100, call_mask=res["call_mask"], capacity=res["capacity"], calls_buf=res["calls_buf"], n_calls=res["n_calls"])
        torch.cuda.synchronize()
        t = sorted(ctx.elapsed_ms(a, b) for a, b in ev)
        b = 16 * P * S + 88 * P
        print(f"P {P} S {S} ({(P + 63) // 64} tiles, {P * S / 1e6:.1f} M records) {'alternating with poisson_call' if alt else 'back to back':30s}: "
              f"median {t[reps // 2] * 1e3:6.1f} us (min {t[0] * 1e3:6.1f}) = {b / t[reps // 2] / 1e9:5.2f} TB/s = {b / t[reps // 2] / 1e9 / 8:5.3f} of peak", flush=True)
    del n, fin

# row stride: config 3's rows are 100 000 records = 1.6 MB apart; does a padded stride (ampli_records.row_stride) change the stream?
P, S = 100_000, 256
import numpy as np
for stride in (100_000, 100_032, 100_096, 100_352, 101_376, 102_400, 100_000):
    buf = torch.zeros((S, stride, 8), dtype=torch.int16, device="cuda")
    dense = cohort(P, S).view(S, P, 8)
    buf[:, :P] = dense
    del dense
    rec = ctx.records(buf, "u16", S, row_stride=stride)
    fin = ctx.error_reduce_records(rec, P, None, finalize=True)
    reps = 30
    ev = [[ctx.event(), ctx.event()] for _ in range(reps)]
    for i in range(reps + 3):
        j = i - 3
        if j >= 0:
            ctx.record(ev[j][0])
        ctx.error_reduce_records(rec, P, None, out=fin, finalize=True)
        if j >= 0:
            ctx.record(ev[j][1])
        ctx.poisson_call(tum, PT, fin_t.thr, ref, 100, call_mask=res["call_mask"], capacity=res["capacity"], calls_buf=res["calls_buf"], n_calls=res["n_calls"])
    torch.cuda.synchronize()
    t = sorted(ctx.elapsed_ms(a, b) for a, b in ev)
    print(f"row stride {stride} records ({stride * 16} B): error_estimate median {t[reps // 2] * 1e3:6.1f} us (min {t[0] * 1e3:6.1f}), alternating with poisson_call", flush=True)
    del buf, rec, fin
