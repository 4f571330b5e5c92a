"""error_estimate on 16-byte records, config 3: a few launches for rocprofv3 --pmc / --kernel-trace (diagnostic builds via AMPLISOLVE_HIP_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch
from amplisolve_amd import Context
P, S = 100_000, 256
c = Context(0)
n = c.synth_fill(P, S, first_sample=0, seed=0xA3F15019, depth=2000)
c.set_record_layout(True)
n16, ok = c.pack16(n)
f = c.error_estimate(n16, P, 0.002, 100)
for _ in range(30):
    c.error_estimate(n16, P, 0.002, 100, out=f)
torch.cuda.synchronize()
