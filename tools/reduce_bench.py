"""error_estimate on config 3 in all three record layouts, measured inside an alternating loop with poisson_call (the way
bench.py's step sees it).  AMPLISOLVE_HIP_LIB selects the library (A/B runs on one box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from amplisolve_amd import Context

P, S, T = 100_000, 256, 96
SEED = 0xA3F15019
torch.cuda.set_stream(torch.cuda.Stream())
base = Context(0)
normals = base.synth_fill(P, S, seed=SEED, depth=2000)
tumours = base.synth_fill(P, T, seed=SEED, depth=2000, tumour=True)
ref_code = base.synth_ref(P, seed=SEED)
for name, rb in [(n, {"u24": 24, "i32": 32, "u16": 16}[n]) for n in (sys.argv[1:] or ["u24", "i32", "u16"])]:
    ctx = Context(0)
    n, _ = ctx.pack(normals, name)
    t, _ = ctx.pack(tumours, name)
    ctx.set_record_layout(name)
    # RB_SPLITS / RB_ROWS / RB_DRAIN: ampli_set_tuning / ampli_set_poisson_tuning knobs for A/B runs (0 = automatic)
    ctx.set_tuning(int(os.environ.get("RB_SPLITS", "0")))
    ctx.set_reduce_compact(int(os.environ.get("RB_COMPACT", "1")))  # 1: the compact-state kernels (uint16 / 24-bit records); 2: uint16 only; 0: general kernel
    ctx.set_poisson_tuning(int(os.environ.get("RB_ROWS", "0")), int(os.environ.get("RB_DRAIN", "0")))
    fin = ctx.error_estimate(n, P, 0.002, 100)
    res = ctx.poisson_call(t, P, fin.thr, ref_code, 100, capacity=1 << 20)
    reps = 30
    evs = [[ctx.event() for _ in range(3)] for _ in range(reps)]
    for _ in range(3):
        ctx.error_estimate(n, P, 0.002, 100, out=fin)
        ctx.poisson_call(t, P, fin.thr, ref_code, 100, call_mask=res["call_mask"], capacity=res["capacity"], calls_buf=res["calls_buf"], n_calls=res["n_calls"])
    for i in range(reps):
        ctx.record(evs[i][0])
        ctx.error_estimate(n, P, 0.002, 100, out=fin)
        ctx.record(evs[i][1])
        ctx.poisson_call(t, P, fin.thr, ref_code, 100, call_mask=res["call_mask"], capacity=res["capacity"], calls_buf=res["calls_buf"], n_calls=res["n_calls"])
        ctx.record(evs[i][2])
    torch.cuda.synchronize()
    red = sorted(ctx.elapsed_ms(e[0], e[1]) for e in evs)
    pc = sorted(ctx.elapsed_ms(e[1], e[2]) for e in evs)
    b = rb * P * S + 88 * P
    print(f"{os.environ.get('RB_TAG', '')} {name}: error_estimate median {red[reps // 2] * 1e3:6.1f} us (min {red[0] * 1e3:6.1f}) = {b / red[reps // 2] / 1e9:5.2f} TB/s = {b / red[reps // 2] / 1e9 / 8:5.3f} of peak;"
          f"  poisson_call median {pc[reps // 2] * 1e3:6.1f} us;  flags {ctx.flags()}", flush=True)
    ctx.close()
    del n, t, fin, res
