"""Does a pass of config 3 get shorter when its positions are split in two ranges on two streams -- a first range of exactly one
round of error_reduce workgroups (1280 tiles at five waves per SIMD), the rest on a second stream -- so that the second range's
kernels and the first range's poisson_call fill the thin last round?  One batch resident in HBM, uint16 records; every range has
its own outputs; results of the two ranges together == the unsplit pass (checked).  usage: python tools/split_probe.py [first_tiles ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

from amplisolve_amd import Context

P, S, T = 100_000, 256, 96
SEED = 0xA3F15019
base = Context(0)
normals = base.synth_fill(P, S, seed=SEED, depth=2000)
tumours = base.synth_fill(P, T, seed=SEED, depth=2000, tumour=True)
ref_code = base.synth_ref(P, seed=SEED)
base.set_record_layout("u16")
n16, _ = base.pack(normals, "u16")
t16, _ = base.pack(tumours, "u16")
n16 = n16.view(S, P, -1)
t16 = t16.view(T, P, -1)
del normals, tumours
cap = 1 << 20


JOIN = os.environ.get("SP_JOIN", "0") == "1"   # both ranges joined at the end of EVERY pass (a library call that forks and joins inside)
PRIO = os.environ.get("SP_PRIO", "0") == "1"   # the first range's stream at high priority (its kernels dispatch first)
_made = [0]


def make_lane(p0, p1):
    st = torch.cuda.Stream(device=0, priority=-1 if (PRIO and _made[0] % 2 == 0 and p0 == 0) else 0)
    _made[0] += 1
    with torch.cuda.stream(st):
        c = Context(0)
        c.set_record_layout("u16")
        Pp = p1 - p0
        nrec = c.records(n16[:, p0:p1], "u16", S, row_stride=P)
        trec = c.records(t16[:, p0:p1], "u16", T, row_stride=P)
        rc = ref_code[p0:p1].contiguous()
        fin = c.error_reduce_records(nrec, Pp, None, finalize=True)
        res = c.poisson_call_records(trec, Pp, fin.thr, rc, 100, capacity=cap)
    return dict(c=c, st=st, Pp=Pp, nrec=nrec, trec=trec, rc=rc, fin=fin, res=res)


def step(l):
    c = l["c"]
    c.error_reduce_records(l["nrec"], l["Pp"], None, out=l["fin"], finalize=True)
    r = l["res"]
    c.poisson_call_records(l["trec"], l["Pp"], l["fin"].thr, l["rc"], 100, call_mask=r["call_mask"], capacity=r["capacity"],
                           calls_buf=r["calls_buf"], n_calls=r["n_calls"])


def timed(lanes, steps=50, warm=5):
    for _ in range(warm):
        for l in lanes:
            step(l)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(torch.cuda.current_stream())
    for l in lanes:  # every lane starts behind the common start event
        l["st"].wait_event(e0)
    cur = torch.cuda.current_stream()
    for _ in range(steps):
        for l in lanes:
            step(l)
        if JOIN and len(lanes) > 1:  # join, then fork again: the next pass starts when every range of this one is done
            for l in lanes:
                cur.wait_stream(l["st"])
            for l in lanes:
                l["st"].wait_stream(cur)
    for l in lanes:
        cur.wait_stream(l["st"])
    e1.record(torch.cuda.current_stream())
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps


whole = make_lane(0, P)
torch.cuda.synchronize()
print(f"join at the end of every pass: {JOIN}; first range at high priority: {PRIO}", flush=True)
for rep in range(int(os.environ.get("SP_REPS", "3"))):
    print(f"unsplit, one stream: {timed([whole]) * 1e3:7.1f} us per pass", flush=True)
    for first in [int(a) for a in sys.argv[1:]] or [1280, 1024, 782]:
        a, b = make_lane(0, first * 64), make_lane(first * 64, P)
        torch.cuda.synchronize()
        ms = timed([a, b])
        same = (torch.equal(torch.cat([a["fin"].thr, b["fin"].thr], dim=2), whole["fin"].thr) and
                torch.equal(torch.cat([a["res"]["call_mask"], b["res"]["call_mask"]], dim=1), whole["res"]["call_mask"]))
        print(f"split at {first} tiles ({first * 64} + {P - first * 64} positions), two streams: {ms * 1e3:7.1f} us per pass   same outputs: {same}", flush=True)
        a["c"].close(); b["c"].close()
    for n in (3, 4, 6):  # equal ranges (tile-aligned) on n streams
        tiles = (P + 63) // 64
        cuts = [min(P, (tiles * k // n) * 64) for k in range(n)] + [P]
        ls = [make_lane(cuts[k], cuts[k + 1]) for k in range(n)]
        torch.cuda.synchronize()
        ms = timed(ls)
        same = (torch.equal(torch.cat([l["fin"].thr for l in ls], dim=2), whole["fin"].thr) and
                torch.equal(torch.cat([l["res"]["call_mask"] for l in ls], dim=1), whole["res"]["call_mask"]))
        print(f"{n} equal ranges on {n} streams: {ms * 1e3:7.1f} us per pass   same outputs: {same}", flush=True)
        for l in ls:
            l["c"].close()
