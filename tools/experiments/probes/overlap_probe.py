"""Do two independent batches on two streams overlap usefully?  (N = 1, config 3)"""
import sys, time, torch
sys.path.insert(0, '.')
from amplisolve_amd import Context
from amplisolve_amd.api import POISSON_PREFILTER
P, S, T = 100_000, 256, 96
base = Context(0)
nor = base.synth_fill(P, S); tum = base.synth_fill(P, T, tumour=True); ref = base.synth_ref(P)
torch.cuda.synchronize()
def make(n):
    cs = []
    for _ in range(n):
        c = Context(0, own_stream=True)
        fin = c.error_estimate(nor, P); res = c.poisson_call(tum, P, fin.thr, ref, 100, mode=POISSON_PREFILTER, capacity=1 << 20)
        c.sync(); cs.append((c, fin, res))
    return cs
def run(cs, steps):
    for i in range(steps):
        c, fin, res = cs[i % len(cs)]
        c.error_estimate(nor, P, out=fin)
        c.poisson_call(tum, P, fin.thr, ref, 100, mode=POISSON_PREFILTER, call_mask=res["call_mask"], capacity=res["capacity"], calls_buf=res["calls_buf"], n_calls=res["n_calls"])
    for c, _, _ in cs: c.sync()
for n in (1, 2, 3, 1, 2, 3):
    cs = make(n)
    run(cs, 10)
    t0 = time.perf_counter(); run(cs, 60); dt = (time.perf_counter() - t0) / 60
    print(f"{n} stream(s): {dt*1e3:.4f} ms per batch  -> {(P*S+P*T)/dt/1e9:.1f} G evals/s")
    for c, _, _ in cs: c.close()
