"""poisson_call launch-shape sweep on config 3 (24-byte records): rows per wave x drain lanes per strand x drain workgroups
per shard, HIP events over back-to-back calls.  Usage: python tools/poisson_sweep.py [quick]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from amplisolve_amd import Context

P, S, T = 100_000, 256, 96
SEED = 0xA3F15019
torch.cuda.set_stream(torch.cuda.Stream())
ctx = Context(0)
normals = ctx.synth_fill(P, S, seed=SEED, depth=2000)
tumours = ctx.synth_fill(P, T, seed=SEED, depth=2000, tumour=True)
ref_code = ctx.synth_ref(P, seed=SEED)
n24, _ = ctx.pack24(normals)
t24, _ = ctx.pack24(tumours)
del normals, tumours
ctx.set_record_layout("u24")
fin = ctx.error_estimate(n24, P, 0.002, 100)
res = ctx.poisson_call(t24, P, fin.thr, ref_code, 100, capacity=1 << 20)
base_mask = res["call_mask"].clone()
n_base = ctx.n_calls_total(res)
BYTES = 24 * P * T + 33 * P + P * T


def timeit(fn, reps=60):
    for _ in range(5):
        fn()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(reps):
        fn()
    ctx.record(e1)
    torch.cuda.synchronize()
    return ctx.elapsed_ms(e0, e1) / reps * 1e3


def call():
    ctx.poisson_call(t24, P, fin.thr, ref_code, 100, call_mask=res["call_mask"], capacity=res["capacity"], calls_buf=res["calls_buf"],
                     n_calls=res["n_calls"])


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
rows_list = [4, 6, 12] if quick else [2, 3, 4, 5, 6, 8, 12, 24]
lanes_list = [0]
print(f"calls per launch {n_base}; algorithmic bytes {BYTES / 1e6:.1f} MB", flush=True)
for rows in rows_list:
    for lanes in lanes_list:
        for blocks in ([32] if quick or rows != 4 else [8, 16, 32, 64]):
            ctx.set_poisson_tuning(rows, blocks)
            t = timeit(call)
            ok = torch.equal(res["call_mask"], base_mask) and ctx.n_calls_total(res) == n_base
            print(f"rows/wave {rows:3d}  drain blocks/shard {blocks:3d}: {t:7.1f} us  {BYTES / t / 1e6:6.2f} TB/s  "
                  f"{BYTES / t / 1e6 / 8:5.3f} of peak  same={ok}", flush=True)
ctx.set_poisson_tuning()
# error_estimate for reference in the same process
f2 = fin
t = timeit(lambda: ctx.error_estimate(n24, P, 0.002, 100, out=f2))
print(f"error_estimate u24: {t:7.1f} us  {(24 * P * S + 88 * P) / t / 1e6:6.2f} TB/s", flush=True)
