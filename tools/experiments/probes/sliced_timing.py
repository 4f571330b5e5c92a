"""Times the three kernels of the position-sliced merge on one GPU (config 3 shapes, world = 8 emulated)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from amplisolve_amd import Context
from amplisolve_amd.dist import slice_geometry

P, S, n = 100_000, 256, 8
torch.cuda.set_stream(torch.cuda.Stream())
ctx = Context(0)
recs = ctx.synth_fill(P, S, first_sample=0, seed=0xA3F15019, depth=2000)
L, sb, gb, bb = slice_geometry(P, n)
sums = torch.zeros(n * 21 * L, dtype=torch.float64, device="cuda")
gm = torch.zeros(n * 8 * L, dtype=torch.float32, device="cuda")
blocks = torch.zeros(n * bb, dtype=torch.uint8, device="cuda")
recv = torch.zeros(n * 8 * L, dtype=torch.float32, device="cuda")
acc = ctx.new_acc(P)
packed = torch.zeros(21 * P, dtype=torch.float64, device="cuda")


def timeit(name, fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(reps):
        fn()
    ctx.record(e1)
    torch.cuda.synchronize()
    print(f"{name:32s} {ctx.elapsed_ms(e0, e1) / reps * 1e3:8.1f} us")


fin = None
timeit("error_estimate (N=1 fused)", lambda: ctx.error_estimate(recs, P, 0.002, 100))
timeit("error_reduce_packed", lambda: ctx.error_reduce_packed(recs, P, acc, packed, 0.002, 100))
timeit("error_reduce_sliced", lambda: ctx.error_reduce_sliced(recs, P, n, sums, gm, 0.002, 100))
timeit("error_finalize_slice", lambda: ctx.error_finalize_slice(P, n, 3, sums[3 * 21 * L:4 * 21 * L], recv, blocks[3 * bb:4 * bb], 0.002, 100))
out = ctx.error_table_unslice(P, n, blocks)
timeit("error_table_unslice", lambda: ctx.error_table_unslice(P, n, blocks, out=out))
