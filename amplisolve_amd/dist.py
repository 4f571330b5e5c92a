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


def slice_geometry(P: int, world: int, slim: bool = False):
    """(L, sums_bytes, gm_bytes, block_bytes) of the position-sliced exchange (ampli_slice_len / ampli_slice_bytes_fmt)."""
    from ._lib import hip_lib

    lib = hip_lib()
    a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
    if lib.ampli_slice_bytes_fmt(P, world, 1 if slim else 0, C.byref(a), C.byref(b), C.byref(c)) != 0:
        raise ValueError("ampli_slice_bytes_fmt failed")
    return int(lib.ampli_slice_len(P, world)), a.value, b.value, c.value


def slice_planes(slim: bool = False) -> int:
    """doubles per position in the sums of the sliced exchange: 21 (wide) or 14 (slim: integer planes packed, include/amplisolve_hip.h)"""
    return 14 if slim else 21


class SlicedMerger:
    """Position-sliced merge of the ranks' partial error statistics (include/amplisolve_hip.h, "Position-sliced merge").

    Rank k owns the positions [k*L, (k+1)*L).  Per batch:

        error_reduce_sliced -> sums[slot] f64 [world][planes][L] (planes = 21, or 14 in the slim format), gm[slot] f32 [world][8][L]
        start_exchange(slot):  reduce-scatter(SUM) sums -> sum_slice[slot] [planes][L]
                               all-to-all gm           -> gm_recv[slot] [world][8][L]
        error_finalize_slice   -> block[slot]          (this rank's slice of the error table)
        start_gather(slot):    all-gather block        -> blocks[slot] [world][block_bytes]
        error_table_unslice    -> the plane-major error table on every rank

    Received bytes per rank and batch at world = 8, P = 100 k: 14.7 + 2.8 + 7.7 = 25.3 MB (slim sums: 9.8 + 2.8 + 7.7 = 20.4 MB),
    against 29.4 + 33.6 = 63 MB for the all-reduce + all-gather of whole tables (TableMerger).  `depth` buffer sets let a caller keep that many
    batches in flight; the collectives ride on RCCL's stream.  Backends without reduce-scatter / all-to-all on device
    tensors (gloo: rehearsals and CPU tests) get the same data movement out of all-reduce / all-gather.
    """

    def __init__(self, P: int, world: int, rank: int, device, group=None, depth: int = 3, batches: int = 1, slim: bool = False):
        """batches = G > 1: every buffer holds G independent batches per slice chunk (ampli_set_slice_group(ctx, G, g)
        selects the batch a kernel call addresses) and ONE round of collectives serves all of them -- fewer, larger
        messages and a third of the cross-stream waits per batch.  slim: the sums travel as 14 packed planes instead of 21
        (the contexts that fill / read these buffers must be in the same format: Context.set_slice_format)."""
        import torch
        import torch.distributed as dist

        self.P, self.world, self.rank, self.group, self.depth, self.batches = P, world, rank, group, depth, batches
        self.slim, self.planes = bool(slim), slice_planes(slim)
        self.L, sums_bytes, gm_bytes, self.block_bytes = slice_geometry(P, world, slim)
        L = self.L * batches  # every plane length below is per slice chunk = `batches` batches back to back
        sums_bytes, gm_bytes = sums_bytes * batches, gm_bytes * batches
        self.native = dist.get_backend(group) == "nccl"
        z = dict(device=device)
        # zeroed once: entries of the padding positions (>= P) are never written by the kernels
        self.sums = [torch.zeros(world * self.planes * L, dtype=torch.float64, **z) for _ in range(depth)]
        self.gm = [torch.zeros(world * 8 * L, dtype=torch.float32, **z) for _ in range(depth)]
        self.sum_slice = [torch.zeros(self.planes * L, dtype=torch.float64, **z) for _ in range(depth)]
        self.gm_recv = [torch.zeros(world * 8 * L, dtype=torch.float32, **z) for _ in range(depth)]
        self.block = [torch.zeros(batches * self.block_bytes, dtype=torch.uint8, **z) for _ in range(depth)]
        self.blocks = [torch.zeros(world * batches * self.block_bytes, dtype=torch.uint8, **z) for _ in range(depth)]
        self._a2a_tmp = None if self.native else torch.zeros(world * world * 8 * L, dtype=torch.float32, **z)
        assert sums_bytes == self.sums[0].numel() * 8 and gm_bytes == self.gm[0].numel() * 4

    def probe(self):
        """The three collectives of the exchange on tiny tensors: raises where the runtime lacks one (it does so at the
        call, identically on every rank), so that all ranks can agree on a fall-back BEFORE the pipeline starts."""
        import torch
        import torch.distributed as dist

        if not self.native:
            return
        dev, w = self.sums[0].device, self.world
        a = torch.zeros(w * 4, dtype=torch.float64, device=dev)
        dist.reduce_scatter_tensor(torch.zeros(4, dtype=torch.float64, device=dev), a, op=dist.ReduceOp.SUM, group=self.group)
        b = torch.zeros(w * 4, dtype=torch.float32, device=dev)
        dist.all_to_all_single(torch.zeros_like(b), b, group=self.group)
        c = torch.zeros(8, dtype=torch.uint8, device=dev)
        dist.all_gather_into_tensor(torch.zeros(w * 8, dtype=torch.uint8, device=dev), c, group=self.group)
        torch.cuda.synchronize(dev)

    def start_exchange(self, slot: int):
        import torch.distributed as dist

        L, w, r = self.L * self.batches, self.world, self.rank
        if self.native:
            h1 = dist.reduce_scatter_tensor(self.sum_slice[slot], self.sums[slot], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            h2 = dist.all_to_all_single(self.gm_recv[slot], self.gm[slot], group=self.group, async_op=True)
            return (h1, h2)
        # rehearsal form: all-reduce + own chunk; all-gather + own column
        dist.all_reduce(self.sums[slot], op=dist.ReduceOp.SUM, group=self.group)
        self.sum_slice[slot].copy_(self.sums[slot][r * self.planes * L:(r + 1) * self.planes * L])
        dist.all_gather_into_tensor(self._a2a_tmp, self.gm[slot], group=self.group)
        self.gm_recv[slot].view(w, 8 * L).copy_(self._a2a_tmp.view(w, w, 8 * L)[:, r, :])
        return ()

    def start_gather(self, slot: int):
        import torch.distributed as dist

        return (dist.all_gather_into_tensor(self.blocks[slot], self.block[slot], group=self.group, async_op=True),)

    @staticmethod
    def wait(handle):
        """The current stream waits for the collectives of `handle`."""
        for h in handle:
            h.wait()

    def bytes_received_per_step(self) -> dict:
        """What one batch's exchange brings to a rank (the collectives are issued once per `batches` batches, so a round
        moves `batches` times this): every other rank's contribution to the own slice, every other rank's block."""
        w, L = self.world, self.L
        rs, a2a, ag = (w - 1) * self.planes * L * 8, (w - 1) * 8 * L * 4, (w - 1) * self.block_bytes
        return {"reduce_scatter_f64": rs, "all_to_all_f32": a2a, "all_gather_u8": ag, "total": rs + a2a + ag,
                "sums_format": f"{'slim' if self.slim else 'wide'}: {self.planes} doubles per position"}

    def time_collectives(self, reps: int = 10) -> dict:
        """Each of the three collectives alone, `reps` rounds back to back on slot 0's buffers, nothing else on the device:
        mean milliseconds per ROUND (a round serves `batches` batches).  Call outside any pipeline: it overwrites slot 0."""
        import torch
        import torch.distributed as dist

        dev = self.sums[0].device
        out = {}
        for name, start in (("exchange(reduce_scatter_f64+all_to_all_f32)", lambda: self.start_exchange(0)), ("all_gather_u8", lambda: self.start_gather(0))):
            self.wait(start())
            torch.cuda.synchronize(dev)
            dist.barrier(self.group)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(reps):
                self.wait(start())
            b.record()
            torch.cuda.synchronize(dev)
            out[name] = a.elapsed_time(b) / reps
        if self.native:
            for name, fn in (("reduce_scatter_f64", lambda: dist.reduce_scatter_tensor(self.sum_slice[0], self.sums[0], op=dist.ReduceOp.SUM, group=self.group)),
                             ("all_to_all_f32", lambda: dist.all_to_all_single(self.gm_recv[0], self.gm[0], group=self.group))):
                fn()
                torch.cuda.synchronize(dev)
                dist.barrier(self.group)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    fn()
                b.record()
                torch.cuda.synchronize(dev)
                out[name] = a.elapsed_time(b) / reps
        return out
