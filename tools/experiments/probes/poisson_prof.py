"""One poisson_call launch shape on config 3 (24-byte records), repeated -- to sit under rocprofv3.
Usage: python3 tools/poisson_prof.py ROWS BLOCKS [REPS]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

from amplisolve_amd import Context

rows, blocks = (int(x) for x in sys.argv[1:3])
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
P, S, T = 100_000, 256, 96
SEED = 0xA3F15019
torch.cuda.set_stream(torch.cuda.Stream())
ctx = Context(0)
normals = ctx.synth_fill(P, S, seed=SEED, depth=2000)
tumours = ctx.synth_fill(P, T, seed=SEED, depth=2000, tumour=True)
ref_code = ctx.synth_ref(P, seed=SEED)
LAYOUT = os.environ.get("PP_LAYOUT", "u24")  # record layout the launches read
n24, _ = ctx.pack(normals, LAYOUT)
t24, _ = ctx.pack(tumours, LAYOUT)
del normals, tumours
ctx.set_record_layout(LAYOUT)
fin = ctx.error_estimate(n24, P, 0.002, 100)
ctx.set_poisson_tuning(rows, blocks)
res = ctx.poisson_call(t24, P, fin.thr, ref_code, 100, capacity=1 << 20)
e0, e1 = ctx.event(), ctx.event()
ctx.record(e0)
for _ in range(reps):
    ctx.poisson_call(t24, P, fin.thr, ref_code, 100, call_mask=res["call_mask"], capacity=res["capacity"], calls_buf=res["calls_buf"],
                     n_calls=res["n_calls"])
ctx.record(e1)
torch.cuda.synchronize()
print(f"rows {rows} blocks {blocks}: {ctx.elapsed_ms(e0, e1) / reps * 1e3:.1f} us per call, {ctx.n_calls_total(res)} calls")
