"""Both record layouts on config 3, interleaved: error_estimate and poisson_call (stream + drain) per layout."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from amplisolve_amd import Context

P, S, T = 100_000, 256, 96
torch.cuda.set_stream(torch.cuda.Stream())
c32, c16, c24 = Context(0), Context(0), Context(0)
c16.set_record_layout("u16")
c24.set_record_layout("u24")
normals = c32.synth_fill(P, S, first_sample=0, seed=0xA3F15019, depth=2000)
tumours = c32.synth_fill(P, T, first_sample=0, seed=0xA3F15019, depth=2000, tumour=True)
ref_code = c32.synth_ref(P, seed=0xA3F15019)
n16, ok1 = c16.pack16(normals)
t16, ok2 = c16.pack16(tumours)
n24, ok3 = c24.pack24(normals)
t24, ok4 = c24.pack24(tumours)
assert ok1 and ok2 and ok3 and ok4
fin = {}
res = {}


def timeit(ctx, fn, reps=40):
    for _ in range(4):
        fn()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(reps):
        fn()
    ctx.record(e1)
    torch.cuda.synchronize()
    return ctx.elapsed_ms(e0, e1) / reps * 1e3


for rnd in range(3):
    for name, ctx, nr, tr in (("i32", c32, normals, tumours), ("u24", c24, n24, t24), ("u16", c16, n16, t16)):
        fin[name] = ctx.error_estimate(nr, P, 0.002, 100)
        res[name] = ctx.poisson_call(tr, P, fin[name].thr, ref_code, 100, capacity=1 << 20)
        f, r = fin[name], res[name]
        t_red = timeit(ctx, lambda: ctx.error_estimate(nr, P, 0.002, 100, out=f))
        t_call = timeit(ctx, lambda: ctx.poisson_call(tr, P, f.thr, ref_code, 100, call_mask=r["call_mask"], capacity=r["capacity"],
                                                      calls_buf=r["calls_buf"], n_calls=r["n_calls"]))

        def step():
            ctx.error_estimate(nr, P, 0.002, 100, out=f)
            ctx.poisson_call(tr, P, f.thr, ref_code, 100, call_mask=r["call_mask"], capacity=r["capacity"], calls_buf=r["calls_buf"], n_calls=r["n_calls"])
        t_step = timeit(ctx, step)
        print(f"{name}: error_estimate {t_red:7.1f} us   poisson_call {t_call:7.1f} us   step {t_step:7.1f} us   "
              f"-> {(P * S + P * T) / (t_step * 1e-6):.3e} evaluations/s", flush=True)
for o in ("u16", "u24"):
    same = all(torch.equal(getattr(fin["i32"], k).view(torch.uint8), getattr(fin[o], k).view(torch.uint8)) for k in ("rate", "thr", "code", "germ_present"))
    print(o, "tables identical:", same, " masks identical:", torch.equal(res["i32"]["call_mask"], res[o]["call_mask"]))
