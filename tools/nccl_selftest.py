"""The three RCCL calls of the position-sliced merge (dist.SlicedMerger, native path) on a real RCCL communicator of
size 1 -- the only size a one-GPU box offers: checks signatures, dtypes (f64 reduce-scatter, f32 all-to-all, u8
all-gather) and the stream ordering with kernels launched through the C ABI.
    python -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29591 tools/nccl_selftest.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from amplisolve_amd import Context
from amplisolve_amd.dist import SlicedMerger

torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
main = torch.cuda.Stream()
torch.cuda.set_stream(main)
ctx = Context(0)
P, S, T = 20_000, 16, 4
normals = ctx.synth_fill(P, S)
tumours = ctx.synth_fill(P, T, tumour=True)
ref_code = ctx.synth_ref(P)
m = SlicedMerger(P, 1, 0, ctx.device)
assert m.native
single = ctx.error_estimate(normals, P)
for slot in range(3):  # the bench's pipeline shape, one batch per slot
    ctx.error_reduce_sliced(normals, P, 1, m.sums[slot], m.gm[slot])
    m.wait(m.start_exchange(slot))
    ctx.error_finalize_slice(P, 1, 0, m.sum_slice[slot], m.gm_recv[slot], m.block[slot])
    m.wait(m.start_gather(slot))
    fin = ctx.error_table_unslice(P, 1, m.blocks[slot])
    a = ctx.poisson_call(tumours, P, m.blocks[slot], ref_code, 100, capacity=1 << 16, blocks_of=1)
    b = ctx.poisson_call(tumours, P, single.thr, ref_code, 100, capacity=1 << 16)
    torch.cuda.synchronize()
    for k in ("rate", "thr", "code", "germ_present"):
        assert torch.equal(getattr(fin, k).view(torch.uint8), getattr(single, k).view(torch.uint8)), k
    assert torch.equal(a["call_mask"], b["call_mask"])
t = torch.tensor([1.5], dtype=torch.float64, device=ctx.device)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
f = torch.tensor([1], dtype=torch.int32, device=ctx.device)
dist.all_reduce(f, op=dist.ReduceOp.MIN)
dist.barrier()
print("nccl selftest ok: reduce_scatter_tensor(f64), all_to_all_single(f32), all_gather_into_tensor(u8), all_reduce MAX/MIN, barrier on", dist.get_backend(), "world", dist.get_world_size())
dist.destroy_process_group()
