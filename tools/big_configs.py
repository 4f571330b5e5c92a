"""BASELINE configs 4 and 5 at FULL size on one GPU (they fit in 288 GB): timing + oracle check on a slice."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from amplisolve_amd import Context
from amplisolve_amd.api import POISSON_PREFILTER
from oracle import pyoracle as orc
layout = sys.argv[1] if len(sys.argv) > 1 else "u24"  # i32 | u24 | u16 (u16 does not fit config 5)
RB = {"i32": 32, "u24": 24, "u16": 16}[layout]
ctx = Context(0)
for name, P, S, T, depth in (("c4 100k x 1024 normals x 1024 tumours", 100_000, 1024, 1024, 2000),
                             ("c5 1M x 256 normals x 64 tumours, 50000x", 1_000_000, 256, 64, 50_000)):
    nor = ctx.synth_fill(P, S, depth=depth); tum = ctx.synth_fill(P, T, depth=depth, tumour=True); ref = ctx.synth_ref(P)
    mask = torch.empty((T, P), dtype=torch.uint8, device="cuda")
    nor32, tum32 = nor, tum
    ctx.set_record_layout("i32")
    if layout != "i32":
        nor, ok1 = ctx.pack(nor32, layout); tum, ok2 = ctx.pack(tum32, layout)
        if not (ok1 and ok2):
            print(f"{name}: does not fit {layout}"); continue
        ctx.set_record_layout(layout)
    fin = ctx.error_estimate(nor, P)
    res = ctx.poisson_call(tum, P, fin.thr, ref, 100, mode=POISSON_PREFILTER, call_mask=mask, capacity=1 << 22)
    torch.cuda.synchronize()
    assert ctx.flags() == 0 and int(fin.flags.item()) == 0
    e = [ctx.event() for _ in range(3)]
    n = 5
    ctx.record(e[0])
    for _ in range(n): ctx.error_estimate(nor, P, out=fin)
    ctx.record(e[1])
    for _ in range(n): ctx.poisson_call(tum, P, fin.thr, ref, 100, mode=POISSON_PREFILTER, call_mask=mask, capacity=1 << 22, calls_buf=res["calls_buf"], n_calls=res["n_calls"])
    ctx.record(e[2])
    t_r, t_c = ctx.elapsed_ms(e[0], e[1]) / n, ctx.elapsed_ms(e[1], e[2]) / n
    sl = slice(P // 3, P // 3 + 400)
    o_acc = orc.error_reduce(nor32[:, sl].cpu().numpy(), 400)
    o_fin = orc.error_finalize(o_acc)
    ok1 = np.array_equal(fin.thr[:, :, sl].cpu().numpy().view(np.int32), o_fin["thr"].view(np.int32))
    o_call = orc.poisson_call(tum32[:16, sl].cpu().numpy(), 400, o_fin["thr"], ref[sl].cpu().numpy(), 100, dense=False)
    ok2 = np.array_equal(mask[:16, sl].cpu().numpy(), o_call["call_mask"])
    print(f"{name} [{layout}]: error_estimate {t_r:.3f} ms ({RB*P*S/t_r/1e6:.0f} GB/s), poisson_call {t_c:.3f} ms ({(RB+1)*P*T/t_c/1e6:.0f} GB/s), "
          f"{(P*S+P*T)/(t_r+t_c)/1e-3/1e9:.1f} G evals/s, calls {ctx.n_calls_total(res)}, oracle slice thr {ok1} mask {ok2}, order_sensitive {o_acc['order_sensitive']}")
    del nor, tum, nor32, tum32, mask, res, fin
    torch.cuda.empty_cache()
