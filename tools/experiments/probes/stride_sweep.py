"""error_estimate on config 3's cohort (100 000 positions x 256 samples, uint16 records) against the row stride of the record array
(ampli_records.row_stride: records between the same position of consecutive samples), alternating with a poisson_call as in the bench
loop.  usage: python tools/stride_sweep.py [first last step]   (records; default 100000 106496 256)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

from amplisolve_amd import Context

SEED = 0xA3F15019
P, S, T = 100_000, 256, 96
a = [int(x) for x in sys.argv[1:4]] + [100_000, 106_496, 256][len(sys.argv[1:4]):]
ctx = Context(0)
ctx.set_record_layout("u16")
dense = ctx.pack(ctx.synth_fill(P, S, seed=SEED, depth=2000), "u16")[0].view(S, P, 8)
tum = ctx.pack(ctx.synth_fill(P, T, seed=SEED, depth=2000, tumour=True), "u16")[0]
ref = ctx.synth_ref(P, seed=SEED)
fin0 = ctx.error_estimate(dense.view(S, -1, 8), P, 0.002, 100)
res = ctx.poisson_call(tum, P, fin0.thr, ref, 100, capacity=1 << 20)
out = []
for stride in list(range(a[0], a[1] + 1, a[2])) + [a[0]]:
    buf = torch.zeros((S, stride, 8), dtype=torch.int16, device="cuda")
    buf[:, :P] = dense
    rec = ctx.records(buf, "u16", S, row_stride=stride)
    fin = ctx.error_reduce_records(rec, P, None, finalize=True)
    reps = 24
    ev = [[ctx.event(), ctx.event()] for _ in range(reps)]
    for i in range(reps + 3):
        j = i - 3
        if j >= 0:
            ctx.record(ev[j][0])
        ctx.error_reduce_records(rec, P, None, out=fin, finalize=True)
        if j >= 0:
            ctx.record(ev[j][1])
        ctx.poisson_call(tum, P, fin0.thr, ref, 100, call_mask=res["call_mask"], capacity=res["capacity"], calls_buf=res["calls_buf"], n_calls=res["n_calls"])
    torch.cuda.synchronize()
    t = sorted(ctx.elapsed_ms(x, y) for x, y in ev)
    same = torch.equal(fin.thr.view(torch.int32), fin0.thr.view(torch.int32))
    print(f"row stride {stride:7d} records = {stride * 16:8d} B (mod 64 KiB {stride * 16 % 65536:6d}, mod 4 KiB {stride * 16 % 4096:5d}): median {t[reps // 2] * 1e3:6.1f} us  min {t[0] * 1e3:6.1f}  same {same}", flush=True)
    del buf, rec, fin
