"""What of poisson_call's fixed cost is the drain's latency chain and what is the launch itself (round 6): config 3, uint16 records,
cold calls (an error_estimate between two of them).  (a) the real thresholds; (b) thresholds of 0.5 everywhere: no record survives the
no-call bound, the queue stays empty, the drain kernel still launches its 1024 workgroups, finds nothing and ends."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from amplisolve_amd import Context

SEED = 0xA3F15019
torch.cuda.set_stream(torch.cuda.Stream())
ctx = Context(0)
ctx.set_record_layout("u16")
P, S, T = 100_000, 256, 96
normals = ctx.pack(ctx.synth_fill(P, S, seed=SEED), "u16")[0]
tum = ctx.pack(ctx.synth_fill(P, T, seed=SEED, tumour=True), "u16")[0]
refc = ctx.synth_ref(P, seed=SEED)
fin = ctx.error_estimate(normals, P, 0.002, 100)
evict = ctx.error_estimate(normals, P, 0.002, 100)
high = torch.full_like(fin.thr, 0.5)
res = ctx.poisson_call(tum, P, fin.thr, refc, 100, capacity=1 << 20)
DRAINS = [int(x) for x in os.environ.get("DF_DRAIN", "0").split(",")]  # drain workgroups per queue shard (0 = the default: 32 for config 3)
for rnd, drain_blocks in [(r, d) for d in DRAINS for r in range(3 if len(DRAINS) == 1 else 2)]:
    ctx.set_poisson_tuning(0, drain_blocks)
    for name, thr in (("real thresholds", fin.thr), ("empty queue    ", high)):
        reps = 40
        evs = [[ctx.event(), ctx.event()] for _ in range(reps)]
        for i in range(reps + 2):
            ctx.error_estimate(normals, P, 0.002, 100, out=evict)
            if i >= 2:
                ctx.record(evs[i - 2][0])
            if os.environ.get("DF_NOLIST"):  # mask only: the drain skips the ballot, the returning atomic and the 64-byte call records
                ctx.poisson_call(tum, P, thr, refc, 100, call_mask=res["call_mask"])
            else:
                ctx.poisson_call(tum, P, thr, refc, 100, call_mask=res["call_mask"], capacity=res["capacity"], calls_buf=res["calls_buf"], n_calls=res["n_calls"])
            if i >= 2:
                ctx.record(evs[i - 2][1])
        torch.cuda.synchronize()
        v = sorted(ctx.elapsed_ms(a, b) for a, b in evs)
        print(f"round {rnd} drain workgroups per shard {drain_blocks:3d} {name}: poisson_call median {v[reps // 2] * 1e3:6.1f} us  min {v[0] * 1e3:6.1f}   calls {ctx.n_calls_total(res)}", flush=True)
ctx.close()
