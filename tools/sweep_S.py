import sys, torch
sys.path.insert(0, '.')
from amplisolve_amd import Context
ctx = Context(0)
P = 100_000
full = ctx.synth_fill(P, 256)
for fused in (True, False):
    for S in (4, 8, 16, 32, 64, 128, 256):
        nor = full[:S].contiguous()
        acc = ctx.new_acc(P); fin = ctx.error_estimate(nor, P)
        fn = (lambda: ctx.error_estimate(nor, P, out=fin)) if fused else (lambda: ctx.error_reduce(nor, P, acc=acc))
        for _ in range(3): fn()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0)
        for _ in range(20): fn()
        ctx.record(e1)
        ms = ctx.elapsed_ms(e0, e1) / 20
        print(f"fused={fused} S={S:4d} {ms*1e3:8.1f} us  {32*P*S/ms/1e6:8.1f} GB/s")
