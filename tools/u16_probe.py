"""Timing experiment: error_estimate on 16-byte (8 x u16) records vs the shipped 32-byte records (config 3)."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

which = sys.argv[1] if len(sys.argv) > 1 else "driver"
if which == "driver":
    for w in ("i32", "u16", "i32", "u16"):
        env = dict(os.environ)
        if w == "u16":
            env["AMPLISOLVE_HIP_LIB"] = os.path.join(ROOT, "_variants", "u16.so")
        subprocess.run([sys.executable, __file__, w], env=env, check=True)
    sys.exit(0)

from amplisolve_amd import Context
P, S = 100_000, 256
torch.cuda.set_stream(torch.cuda.Stream())
ctx = Context(0)
recs = ctx.synth_fill(P, S, first_sample=0, seed=0xA3F15019, depth=2000)
if which == "u16":
    absent = recs[..., 0] == -(2 ** 31)
    r16 = recs.to(torch.int16)
    r16[..., 0][absent] = -1
    data = r16.contiguous().view(torch.int32)
    ref = torch.load("/tmp/u16_ref.pt")
else:
    data = recs
fin = ctx._new_error_table(P)


def run():
    rc = ctx.lib.ampli_error_estimate(ctx.h, data.data_ptr(), P, 0, None, S, 0.002, 100, None, fin.rate.data_ptr(), fin.code.data_ptr(),
                                      fin.thr.data_ptr(), fin.germ_val.data_ptr(), fin.germ_present.data_ptr(), fin.flags.data_ptr())
    assert rc == 0


for _ in range(5):
    run()
e0, e1 = ctx.event(), ctx.event()
ctx.record(e0)
for _ in range(50):
    run()
ctx.record(e1)
torch.cuda.synchronize()
print(which, f"{ctx.elapsed_ms(e0, e1) / 50 * 1e3:.1f} us", flush=True)
if which == "i32":
    torch.save({k: getattr(fin, k).cpu() for k in ("rate", "thr", "code", "germ_val", "germ_present")}, "/tmp/u16_ref.pt")
else:
    ok = all(torch.equal(getattr(fin, k).cpu().view(torch.uint8), ref[k].view(torch.uint8)) for k in ("rate", "thr", "code", "germ_present"))
    print("u16 results == i32 results:", ok, flush=True)
