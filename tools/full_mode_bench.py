"""All-scores mode (AMPLI_POISSON_FULL) of poisson_call on config 3, uint16 records: HIP events around 5 calls, mask compared with the
prefilter mode's.  AMPLISOLVE_HIP_LIB selects the library (A/B runs of variants on one box)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from amplisolve_amd import Context
from amplisolve_amd.api import POISSON_FULL

P, S, T = 100_000, 256, 96
SEED = 0xA3F15019
ctx = Context(0)
normals = ctx.synth_fill(P, S, seed=SEED, depth=2000)
tumours = ctx.synth_fill(P, T, seed=SEED, depth=2000, tumour=True)
ref_code = ctx.synth_ref(P, seed=SEED)
n, _ = ctx.pack(normals, "u16")
t, _ = ctx.pack(tumours, "u16")
ctx.set_record_layout("u16")
fin = ctx.error_estimate(n, P, 0.002, 100)
pre = ctx.poisson_call(t, P, fin.thr, ref_code, 100, capacity=1 << 20)
full = ctx.poisson_call(t, P, fin.thr, ref_code, 100, mode=POISSON_FULL, capacity=1 << 20)
torch.cuda.synchronize()
same = torch.equal(pre["call_mask"], full["call_mask"]) and ctx.n_calls_total(pre) == ctx.n_calls_total(full)
for rep in range(3):
    a, b = ctx.event(), ctx.event()
    ctx.record(a)
    for _ in range(5):
        ctx.poisson_call(t, P, fin.thr, ref_code, 100, mode=POISSON_FULL, call_mask=full["call_mask"])
    ctx.record(b)
    print(f"{os.environ.get('RB_TAG', '')} all-scores mode: {ctx.elapsed_ms(a, b) / 5:.3f} ms per call; mask and call count == prefilter mode's: {same}; flags {ctx.flags()}", flush=True)
