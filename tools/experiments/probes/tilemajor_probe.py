"""TIMING ONLY: error_estimate of config 3 through the red_tilemajor* variants of tools/build_variant.py (they read a tile-major
arrangement out of the same buffer: wrong results, same work).  The buffer holds four samples more than the kernel is told about,
so that the padded tile stride stays inside the allocation.  AMPLISOLVE_HIP_LIB selects the library."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

from amplisolve_amd import Context

P, S = 100_000, 256
torch.cuda.set_stream(torch.cuda.Stream())
for name, rb in (("u16", 16), ("u24", 24), ("i32", 32)):
    ctx = Context(0)
    big = ctx.synth_fill(P, S + 4, seed=0xA3F15019, depth=2000)  # slack: 1563 tiles x (S*64 + 192) records < (S + 4) * P
    if name != "i32":
        big, _ = ctx.pack(big, name)
    n = big[:S] if big.dim() > 1 else big[: S * P * rb]
    ctx.set_record_layout(name)
    fin = ctx.error_estimate(n, P, 0.002, 100)
    other = torch.empty(300 * 1024 * 1024, dtype=torch.uint8, device=ctx.device)
    reps = 30
    evs = [[ctx.event(), ctx.event()] for _ in range(reps)]
    for i in range(reps + 3):
        other.add_(1)  # push the cohort out of the Infinity Cache between passes, like the other half of a step does
        if i >= 3:
            ctx.record(evs[i - 3][0])
        ctx.error_estimate(n, P, 0.002, 100, out=fin)
        if i >= 3:
            ctx.record(evs[i - 3][1])
    torch.cuda.synchronize()
    t = sorted(ctx.elapsed_ms(a, b) for a, b in evs)
    b = rb * P * S + 88 * P
    print(f"{os.environ.get('RB_TAG', '')} {name}: error_estimate median {t[reps // 2] * 1e3:6.1f} us (min {t[0] * 1e3:6.1f}) = {b / t[reps // 2] / 1e9:5.2f} TB/s", flush=True)
    ctx.flags()
    ctx.close()
    del big, n, fin, other
