"""Config 3 as a two-stage pipeline over back-to-back passes: every error_estimate (whole launch) on one stream, every poisson_call
(whole launches) on another, poisson_call k behind error_estimate k, error_estimate k + 2 behind poisson_call k (two error tables used
alternately) -- so that pass k's poisson_call runs under pass k + 1's error_reduce.  Against the one-stream pass and against position
ranges (ampli_set_ranges).  uint16 records.  usage: python tools/role_pipeline_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

from amplisolve_amd import Context

P, S, T = 100_000, 256, 96
SEED = 0xA3F15019
base = Context(0)
base.set_record_layout("u16")
n16 = base.pack(base.synth_fill(P, S, seed=SEED, depth=2000), "u16")[0]
t16 = base.pack(base.synth_fill(P, T, seed=SEED, depth=2000, tumour=True), "u16")[0]
ref = base.synth_ref(P, seed=SEED)
cap = 1 << 20


def overlapping_pair():
    """two streams that really run concurrently (HIP deals streams to a few hardware queues; two on one queue run in turn)"""
    pool = [torch.cuda.Stream() for _ in range(8)]

    def both(a, b):
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        torch.cuda.synchronize()
        e0.record(a)
        b.wait_event(e0)
        with torch.cuda.stream(a):
            torch.cuda._sleep(300_000)
        with torch.cuda.stream(b):
            torch.cuda._sleep(300_000)
        e1.record(a)
        e2.record(b)
        torch.cuda.synchronize()
        return max(e0.elapsed_time(e1), e0.elapsed_time(e2))

    with torch.cuda.stream(pool[0]):
        torch.cuda._sleep(300_000)
    torch.cuda.synchronize()
    ea, eb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ea.record(pool[0])
    with torch.cuda.stream(pool[0]):
        torch.cuda._sleep(300_000)
    eb.record(pool[0])
    torch.cuda.synchronize()
    alone = ea.elapsed_time(eb)
    for i in range(8):
        for j in range(i + 1, 8):
            both(pool[i], pool[j])
            t = both(pool[i], pool[j])
            if t < 1.5 * alone:
                print(f"streams {i} and {j} overlap ({t:.3f} ms for two sleeps of {alone:.3f} ms)", flush=True)
                return pool[i], pool[j]
    print("no overlapping pair found", flush=True)
    return pool[0], pool[1]


sR, sP = overlapping_pair()
with torch.cuda.stream(sR):
    cR = Context(0)
    cR.set_record_layout("u16")
    fins = [cR.error_estimate(n16, P, 0.002, 100) for _ in range(2)]
with torch.cuda.stream(sP):
    cP = Context(0)
    cP.set_record_layout("u16")
    res = cP.poisson_call(t16, P, fins[0].thr, ref, 100, capacity=cap)
torch.cuda.synchronize()


def one_stream(steps):
    for k in range(steps):
        f = base.error_estimate(n16, P, 0.002, 100, out=fins[k & 1])
        base.poisson_call(t16, P, f.thr, ref, 100, call_mask=res["call_mask"], capacity=res["capacity"], calls_buf=res["calls_buf"], n_calls=res["n_calls"])


def pipelined(steps):
    evR = [torch.cuda.Event() for _ in range(steps)]
    evP = [torch.cuda.Event() for _ in range(steps)]
    for k in range(steps):
        if k >= 2:
            sR.wait_event(evP[k - 2])  # the table this launch overwrites was read by poisson_call k - 2
        cR.error_estimate(n16, P, 0.002, 100, out=fins[k & 1])
        evR[k].record(sR)
        sP.wait_event(evR[k])
        cP.poisson_call(t16, P, fins[k & 1].thr, ref, 100, call_mask=res["call_mask"], capacity=res["capacity"], calls_buf=res["calls_buf"], n_calls=res["n_calls"])
        evP[k].record(sP)


def timed(fn, steps=50, warm=5, streams=()):
    fn(warm)
    torch.cuda.synchronize()
    cur = torch.cuda.current_stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(cur)
    for s in streams:
        s.wait_event(e0)
    fn(steps)
    for s in streams:
        cur.wait_stream(s)
    e1.record(cur)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


want_mask = None
for rep in range(3):
    a = timed(one_stream)
    m1 = res["call_mask"].clone()
    b = timed(pipelined, streams=(sR, sP))
    same = torch.equal(m1, res["call_mask"])
    print(f"one stream {a * 1e3:7.1f} us   role pipeline (reduce stream | call stream) {b * 1e3:7.1f} us (same mask: {same})", flush=True)
