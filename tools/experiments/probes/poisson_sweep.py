"""poisson_call launch-shape sweep on config 3 (24-byte records), measured the way bench.py's loop sees it: every call
follows an error_estimate over the 614 MB normal panel, so neither the tumour records nor the thresholds are left in the
Infinity Cache (back-to-back calls read the 230 MB tumour array from the 256 MiB cache and look ~20 % faster than they are).
Usage: python tools/poisson_sweep.py [quick]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

from amplisolve_amd import Context

P, S, T = 100_000, 256, 96
SEED = 0xA3F15019
torch.cuda.set_stream(torch.cuda.Stream())
ctx = Context(0)
normals = ctx.synth_fill(P, S, seed=SEED, depth=2000)
tumours = ctx.synth_fill(P, T, seed=SEED, depth=2000, tumour=True)
ref_code = ctx.synth_ref(P, seed=SEED)
LAYOUT = os.environ.get("SWEEP_LAYOUT", "u24")  # u24 | u16 | i32
n24, _ = (normals, True) if LAYOUT == "i32" else ctx.pack(normals, LAYOUT)
t24, _ = (tumours, True) if LAYOUT == "i32" else ctx.pack(tumours, LAYOUT)
del normals, tumours
ctx.set_record_layout(LAYOUT)
fin = ctx.error_estimate(n24, P, 0.002, 100)
res = ctx.poisson_call(t24, P, fin.thr, ref_code, 100, capacity=1 << 20)
base_mask = res["call_mask"].clone()
n_base = ctx.n_calls_total(res)
BYTES = {'u24': 24, 'u16': 16, 'i32': 32}[LAYOUT] * P * T + 33 * P + P * T


def call():
    ctx.poisson_call(t24, P, fin.thr, ref_code, 100, call_mask=res["call_mask"], capacity=res["capacity"], calls_buf=res["calls_buf"],
                     n_calls=res["n_calls"])


def in_loop(reps=30):
    """mean duration of poisson_call (stream + drain) and of error_estimate when the two alternate"""
    evs = [[ctx.event() for _ in range(3)] for _ in range(reps)]
    for _ in range(3):
        ctx.error_estimate(n24, P, 0.002, 100, out=fin)
        call()
    for i in range(reps):
        ctx.record(evs[i][0])
        ctx.error_estimate(n24, P, 0.002, 100, out=fin)
        ctx.record(evs[i][1])
        call()
        ctx.record(evs[i][2])
    torch.cuda.synchronize()
    red = sum(ctx.elapsed_ms(e[0], e[1]) for e in evs) / reps * 1e3
    pc = sum(ctx.elapsed_ms(e[1], e[2]) for e in evs) / reps * 1e3
    return red, pc


def back_to_back(reps=40):
    for _ in range(4):
        call()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(reps):
        call()
    ctx.record(e1)
    torch.cuda.synchronize()
    return ctx.elapsed_ms(e0, e1) / reps * 1e3


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
print(f"calls per launch {n_base}; algorithmic bytes {BYTES / 1e6:.1f} MB", flush=True)
for rows in ([int(x) for x in os.environ["SWEEP_ROWS"].split(",")] if os.environ.get("SWEEP_ROWS") else [4, 8, 24] if quick else [2, 3, 4, 5, 6, 8, 12, 16, 24]):
  for drain in ([int(x) for x in os.environ["SWEEP_DRAIN"].split(",")] if os.environ.get("SWEEP_DRAIN") else [0]):
    ctx.set_poisson_tuning(rows, drain)
    red, pc = in_loop()
    bb = back_to_back()
    ok = torch.equal(res["call_mask"], base_mask) and ctx.n_calls_total(res) == n_base
    print(f"rows/wave {rows:3d} drain blocks/shard {drain:3d}: in the loop {pc:6.1f} us = {BYTES / pc / 1e6:5.2f} TB/s = {BYTES / pc / 1e6 / 8:5.3f} of peak   "
          f"(back to back {bb:6.1f} us; error_estimate beside it {red:6.1f} us)   same={ok}", flush=True)
ctx.set_poisson_tuning()
