"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The path shards naturally: tumour samples are independent given
the error table, and the normal panel shards by contiguous sample ranges.  The
only exchange step is the merge of the per-position accumulator table:

  * planes that merge by addition (snt f64, srd i64, cnt/nrec i32) -> all-reduce SUM per dtype
    (the double sums are exact inside the envelope, so the reduction order is immaterial);
  * the germ-max triple is order dependent -> all-gather of the gm region + ordered fold on every rank
    (ampli_gm_merge), which reproduces the reference's sequential state machine bit for bit.

The same function runs over gloo on CPU tensors (tests, world_size 2) with a host-side fold.
"""
from __future__ import annotations

import ctypes as C


def shard_range(n: int, rank: int, world: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of n items for `rank`; earlier ranks take the remainder."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def table_regions(P: int):
    """(sum_bytes, gm_offset, gm_bytes) of an accumulator-table buffer (ampli_acc_regions; needs no GPU)."""
    from ._lib import hip_lib

    a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    if hip_lib().ampli_acc_regions(P, C.byref(a), C.byref(b), C.byref(c)) != 0:
        raise ValueError("ampli_acc_regions failed")
    return a.value, b.value, c.value


def merge_error_table(acc, fold, group=None, gather_buf=None):
    """In-place merge of every rank's partial table `acc` (an api.Acc; rank order = sample order).

    fold(acc, gathered_regions, world) folds the gathered germ-max regions into acc's gm planes:
    Context.gm_merge on the GPU, a host fold in the CPU tests."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if world == 1:
        return acc
    _, gm_off, gm_bytes = table_regions(acc.P)
    region = acc.buf[gm_off: gm_off + gm_bytes]
    if gather_buf is None:
        gather_buf = torch.empty(world * gm_bytes, dtype=torch.uint8, device=acc.buf.device)
    # the gm region goes first: it carries the per-shard gm_n the fold needs
    work = [dist.all_gather_into_tensor(gather_buf, region, group=group, async_op=True)]
    work += [dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group, async_op=True) for t in (acc.snt, acc.srd, acc.cnt, acc.nrec)]
    for w in work:
        w.wait()
    fold(acc, gather_buf, world)
    return acc


class TableMerger:
    """Asynchronous, double-buffered form of merge_error_table for a pipeline of independent batches.

    start(acc, slot) packs the additive planes of `acc` into ONE float64 buffer (srd/cnt/nrec are exact in a
    double: < 2^53) and enqueues a single all-reduce plus the all-gather of the germ-max region; finish(...)
    makes the compute stream wait for them, unpacks and folds.  Between the two calls the caller is free to run
    the previous batch's finalize + poisson_call and the next batch's error_reduce: the collectives ride on
    RCCL's own stream over xGMI while the CUs keep streaming HBM.
    """

    def __init__(self, P: int, world: int, device, fold, group=None, pack=None, unpack=None):
        """pack(acc, buf) / unpack(buf, acc): Context.acc_pack / acc_unpack on the GPU (one kernel each);
        default: torch copies (CPU tests)."""
        import torch

        self.P, self.world, self.group, self.fold = P, world, group, fold
        self.pack, self.unpack = pack, unpack
        _, self.gm_off, self.gm_bytes = table_regions(P)
        self.packed = [torch.empty(21 * P, dtype=torch.float64, device=device) for _ in range(2)]
        self.gathered = [torch.empty(world * self.gm_bytes, dtype=torch.uint8, device=device) for _ in range(2)]

    def start(self, acc, slot: int, prepacked: bool = False):
        """prepacked: packed[slot] was already filled by ampli_error_reduce_packed (no pack launch)."""
        import torch.distributed as dist

        P, pk = self.P, self.packed[slot]
        if prepacked:
            pass
        elif self.pack is not None:
            self.pack(acc, pk)
        else:
            pk[0:8 * P].copy_(acc.snt.view(-1))
            pk[8 * P:16 * P].copy_(acc.srd.view(-1))
            pk[16 * P:20 * P].copy_(acc.cnt.view(-1))
            pk[20 * P:21 * P].copy_(acc.nrec)
        # the all-reduce is the critical one (the thresholds hang on it): issue it first
        w_r = dist.all_reduce(pk, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        w_g = dist.all_gather_into_tensor(self.gathered[slot], acc.buf[self.gm_off: self.gm_off + self.gm_bytes],
                                          group=self.group, async_op=True)
        return (w_r, w_g)

    def wait(self, handle):
        """Compute stream waits for both collectives; packed[slot] / gathered[slot] are then ready for
        ampli_error_finalize_merged (no unpack, no table update)."""
        for h in handle:
            h.wait()

    def finish(self, acc, slot: int, handle):
        P, pk = self.P, self.packed[slot]
        w_r, w_g = handle
        w_r.wait()
        if self.unpack is not None:
            self.unpack(pk, acc)
        else:
            acc.snt.view(-1).copy_(pk[0:8 * P])
            acc.srd.view(-1).copy_(pk[8 * P:16 * P])
            acc.cnt.view(-1).copy_(pk[16 * P:20 * P])
            acc.nrec.copy_(pk[20 * P:21 * P])
        w_g.wait()
        self.fold(acc, self.gathered[slot], self.world)
        return acc
