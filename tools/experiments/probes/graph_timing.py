"""A config-2 pass (10 k positions x 32 normals x 8 tumours: launch-bound) enqueued eagerly and replayed as a captured hipGraph.
Timing only -- the bit-equality of the replay is tests/test_gpu_parity.py::test_hipgraph_capture_replays_the_same_pass.
Usage: python tools/graph_timing.py [passes]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from amplisolve_amd import Context
from amplisolve_amd.api import POISSON_PREFILTER

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = Context(0, own_stream=True)
P, S, T = 10_000, 32, 8
normals = g.synth_fill(P, S)
tum = g.synth_fill(P, T, tumour=True)
refc = g.synth_ref(P)
fin = g.error_estimate(normals, P)
res = g.poisson_call(tum, P, fin.thr, refc, 100, mode=POISSON_PREFILTER, capacity=1 << 16)
g.sync()


def one_pass():
    g.error_estimate(normals, P, out=fin)
    g.poisson_call(tum, P, fin.thr, refc, 100, mode=POISSON_PREFILTER, call_mask=res["call_mask"], capacity=res["capacity"],
                   calls_buf=res["calls_buf"], n_calls=res["n_calls"])


g.graph_begin()
one_pass()
graph = g.graph_end()
for fn, name in ((one_pass, "eager"), (lambda: g.graph_launch(graph), "hipGraph replay")):
    fn()
    g.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    g.sync()
    print(f"config-2 pass, {name}: {(time.perf_counter() - t0) / n * 1e6:.1f} us")
g.graph_destroy(graph)
g.close()
