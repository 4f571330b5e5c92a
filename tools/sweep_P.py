import sys, torch
sys.path.insert(0, '.')
from amplisolve_amd import Context
ctx = Context(0)
S, T = 256, 96
for P in (32768, 65536, 98304, 100000, 131072, 196608, 262144):
    nor = ctx.synth_fill(P, S); tum = ctx.synth_fill(P, T, tumour=True); ref = ctx.synth_ref(P)
    acc = ctx.new_acc(P); fin = ctx.error_finalize(ctx.error_reduce(nor, P, acc=acc))
    mask = torch.empty((T, P), dtype=torch.uint8, device="cuda")
    for name, fn, nbytes in (("reduce", lambda: ctx.error_reduce(nor, P, acc=acc), 32 * P * S),
                             ("poisson", lambda: ctx.poisson_call(tum, P, fin.thr, ref, 100, call_mask=mask), 33 * P * T)):
        for _ in range(3): fn()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(20): fn()
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1) / 20
        print(f"P={P:7d} {name:8s} {ms*1e3:8.1f} us  {nbytes/ms/1e6:8.1f} GB/s  tiles={-(-P//64)}")
    del nor, tum
