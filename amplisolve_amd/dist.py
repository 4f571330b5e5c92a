"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The path shards naturally: tumour samples are independent given
the error table, and the normal panel shards by contiguous sample ranges.  The
only exchange step is the merge of the per-position accumulator table:

  * planes that merge by addition (snt f64, srd i64, cnt/nrec/gm_n i32) -> all-reduce SUM per dtype
    (the double sums are exact inside the envelope, so the reduction order is immaterial);
  * the germ-max triple is order dependent -> all-gather of the gm region + ordered fold on every rank
    (ampli_gm_merge), which reproduces the reference's sequential state machine bit for bit.
"""
from __future__ import annotations


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of n items for `rank`; earlier ranks take the remainder."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def merge_error_table(ctx, acc, group=None, gather_buf=None):
    """In-place merge of every rank's partial table `acc` (rank order = sample order)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return acc
    sum_bytes, gm_off, gm_bytes = ctx.regions(acc.P)
    region = acc.buf[gm_off: gm_off + gm_bytes]
    if gather_buf is None:
        gather_buf = torch.empty(world * gm_bytes, dtype=torch.uint8, device=acc.buf.device)
    # gm region first: it contains the per-shard gm_n the fold needs
    w_g = dist.all_gather_into_tensor(gather_buf, region, group=group, async_op=True)
    w_g.wait()
    works = [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True)
             for t in (acc.snt, acc.srd, acc.cnt, acc.nrec)]
    for w in works:
        w.wait()
    ctx.gm_merge(acc, gather_buf, world)
    return acc
