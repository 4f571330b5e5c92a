"""One process per GPU front end of the two command lines.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        -m amplisolve_amd.multi AmpliSolveErrorEstimation panel_design=<bed> reference_genome=<fa> germline_dir=<dir> \\
                                C_value=<f> coverage_cutoff=<i> default_error=<f> output_dir=<dir>
    python -m torch.distributed.run ... -m amplisolve_amd.multi AmpliSolveVariantCalling errorFile=<table> tumour_dir=<dir> \\
                                output_dir=<dir> coverage_cutoff=<i> p_value=<f>

Same tokens, same files, byte-identical outputs as the one-process executables (amplisolve_amd/bin): the C++ host runs
the same pipeline on its shard of the sample files (contiguous ranges of the reference's visit order) and calls back
here for the exchange steps -- the position-sliced merge of the error statistics (dist.SlicedMerger: RCCL
reduce-scatter + all-to-all + all-gather over xGMI) and, for variant calling, a row count prefix + a barrier (tumour
files need no data exchange).  Shard 0 writes the shared files; output_dir must be visible to every process.
AMPLISOLVE_DIST_BACKEND=gloo rehearses the same path where each rank cannot have its own GPU.
Exit status: 0 on success, 1 on failure (unlike the reference, so that the launcher tears the other ranks down).
"""
from __future__ import annotations

import ctypes as C
import os
import sys
import traceback


def _token(arg: str, key: str) -> str:
    """sscanf(arg, "key=%s") as the reference's mains do (EE:300-326, VC:242-260)."""
    if not arg.startswith(key + "="):
        return ""
    rest = arg[len(key) + 1:].lstrip(" \t\n")
    return rest.split()[0] if rest.split() else ""


class Hooks:
    """The collective callbacks of ampli_host_shard over torch.distributed."""

    def __init__(self, rank: int, world: int, device):
        import torch.distributed as dist

        from ._lib import HostShard

        self.rank, self.world, self.device = rank, world, device
        self.native = dist.get_backend() == "nccl"
        self.merger = None
        self.error = None
        self._keep = [HostShard.EE_BUFFERS(self._guard(self.ee_buffers)), HostShard.HOOK(self._guard(self.ee_exchange)),
                      HostShard.HOOK(self._guard(self.ee_gather)), HostShard.OR_FLAGS(self._guard(self.or_flags)),
                      HostShard.ROWS_BEFORE(self._guard(self.rows_before)), HostShard.HOOK(self._guard(self.barrier))]
        self.struct = HostShard(rank, world, None, *self._keep)

    def _guard(self, fn):
        def wrapped(*a):
            try:
                fn(*a)
                return 0
            except Exception:  # noqa: BLE001 -- must not unwind into C++
                self.error = traceback.format_exc()
                return -1
        return wrapped

    def _small(self, values, dtype):
        import torch

        return torch.tensor(values, dtype=dtype, device=self.device if self.native else "cpu")

    # ---- error estimation ----
    def ee_buffers(self, _user, P, out):
        from .dist import SlicedMerger

        m = self.merger = SlicedMerger(int(P), self.world, self.rank, self.device, depth=1)
        for i, t in enumerate((m.sums[0], m.gm[0], m.sum_slice[0], m.gm_recv[0], m.block[0], m.blocks[0])):
            out[i] = t.data_ptr()

    def ee_exchange(self, _user):
        self.merger.wait(self.merger.start_exchange(0))

    def ee_gather(self, _user):
        self.merger.wait(self.merger.start_gather(0))

    def or_flags(self, _user, flags):
        import torch
        import torch.distributed as dist

        bits = self._small([(flags[0] >> b) & 1 for b in range(31)], torch.int32)
        dist.all_reduce(bits, op=dist.ReduceOp.SUM)
        flags[0] = sum(1 << b for b, v in enumerate(bits.tolist()) if v)

    # ---- variant calling ----
    def rows_before(self, _user, mine, before):
        import torch
        import torch.distributed as dist

        counts = self._small([0] * self.world, torch.int64)
        dist.all_gather_into_tensor(counts, self._small([int(mine)], torch.int64))
        before[0] = int(sum(counts.tolist()[: self.rank]))

    def barrier(self, _user):
        import torch
        import torch.distributed as dist

        torch.cuda.synchronize()
        dist.barrier()


def main(argv=None) -> int:
    argv = list(sys.argv if argv is None else argv)
    progs = {"AmpliSolveErrorEstimation": (("panel_design", "reference_genome", "germline_dir", "C_value", "coverage_cutoff", "default_error", "output_dir"),
                                           "ampli_host_run_error_estimation_sharded"),
             "AmpliSolveVariantCalling": (("errorFile", "tumour_dir", "output_dir", "coverage_cutoff", "p_value"),
                                          "ampli_host_run_variant_calling_sharded")}
    if len(argv) < 2 or argv[1] not in progs or len(argv) != 2 + len(progs[argv[1]][0]):
        print("usage: python -m torch.distributed.run --nproc-per-node N -m amplisolve_amd.multi "
              "{AmpliSolveErrorEstimation <7 key=value tokens> | AmpliSolveVariantCalling <5 key=value tokens>}\n"
              "       (tokens and their order as for the one-process executables)")
        return 1
    keys, entry = progs[argv[1]]
    toks = [_token(a, k).encode() for a, k in zip(argv[2:], keys)]

    import torch
    import torch.distributed as dist

    from ._lib import host_lib

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if not torch.cuda.is_available():
        print("amplisolve_amd.multi: no MI355X visible; there is no CPU fallback")
        return 1
    dev = local % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    os.environ["AMPLISOLVE_DEVICE"] = str(dev)  # the C++ host opens its context on this device's default stream
    if world == 1:  # started without a launcher: the one-process pipeline, no process group
        lib = host_lib()
        if entry.startswith("ampli_host_run_error"):
            ref = os.environ.get("AMPLISOLVE_REFBASES_FILE")
            rc = lib.ampli_host_run_error_estimation(*toks, ref.encode() if ref else None)
        else:
            rc = lib.ampli_host_run_variant_calling(*toks)
        return 0 if rc == 0 else 1
    quiet = None
    if rank != 0 and not os.environ.get("AMPLISOLVE_ALL_RANKS_VERBOSE"):
        # the C++ pipeline narrates on stdout like the reference; one narrator is enough -- the others' text is kept
        # in an unnamed file and shown only if that shard fails
        import tempfile

        quiet = tempfile.TemporaryFile()
        sys.stdout.flush()
        os.dup2(quiet.fileno(), 1)
    backend = os.environ.get("AMPLISOLVE_DIST_BACKEND", "nccl")
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend)
    hooks = Hooks(rank, world, torch.device("cuda", dev))
    lib = host_lib()
    try:
        if entry.startswith("ampli_host_run_error"):
            ref = os.environ.get("AMPLISOLVE_REFBASES_FILE")
            rc = getattr(lib, entry)(*toks, ref.encode() if ref else None, C.byref(hooks.struct))
        else:
            rc = getattr(lib, entry)(*toks, C.byref(hooks.struct))
    except Exception:  # noqa: BLE001
        traceback.print_exc()
        rc = -1
    if hooks.error:
        print(hooks.error, file=sys.stderr)
    if rc != 0:
        # the other shards may be waiting in a collective this one will never join: leave at once, without the
        # process-group teardown (which can block on them), so that the launcher ends the whole job
        if quiet is not None:
            quiet.seek(0)
            sys.stderr.write(quiet.read().decode(errors="replace")[-4000:])
        print(f"amplisolve_amd.multi: shard {rank} failed (code {rc})", file=sys.stderr)
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
