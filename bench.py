#!/usr/bin/env python3
"""bench.py -- AmpliSolve hot path on MI355X: error estimation + Poisson calling.

One "step" = one pass of the hot path over one synthetic batch that is already
resident in HBM: error_reduce over the rank's normal-sample shard -> (N>1: RCCL
merge of the accumulator table) -> error_finalize -> poisson_call over the
rank's tumour shard.  Workload at N=1: BASELINE.json configs[2], the one the
metric's target is quoted on (100k positions x 256 normals x 96 tumours);
weak scaling: every rank owns a shard of that size.

Prints ONE JSON line on rank 0 (contract in the task prompt).  `value` counts
position-evaluations: one (position, sample) record pushed through its half of
the path (normals through the gated reduction, tumours through the Poisson
test), whole job, per second.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
SEED = 0xA3F15017 + 2   # seed base + config index

CONFIGS = {
    "c2": dict(P=10_000, S=32, T=8, depth=2000, name="synthetic 10k positions x 32 normals x 8 tumours"),
    "c3": dict(P=100_000, S=256, T=96, depth=2000, name="synthetic 100k positions x 256 normals x 96 tumours (ctDNA-scale)"),
    "c4r": dict(P=100_000, S=128, T=128, depth=2000, name="synthetic 100k x 1024 normals x 1024 tumours, per-rank shard of 8"),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="c3", choices=sorted(CONFIGS))
    ap.add_argument("--mode", default="prefilter", choices=["prefilter", "full"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--splits", type=int, default=0, help="error_reduce sample splits (0 = auto)")
    ap.add_argument("--groups", type=int, default=0, help="error_reduce lane groups per wave (0 = auto, 1, 2, 4)")
    ap.add_argument("--event-every", type=int, default=5, help="record the per-kernel HIP events on every n-th timed step (each record costs a few us of stream time)")
    ap.add_argument("--streams", type=int, default=1, help="N = 1 only: independent batches round-robin over this many HIP streams "
                    "(cross-batch overlap; per-kernel times then include contention, so the default stays 1)")
    ap.add_argument("--async-drain", action="store_true", help="poisson_call's drain kernel on a side stream (measured: no gain on config 3)")
    ap.add_argument("--records", default="auto", choices=["auto", "i32", "u24", "u16"],
                    help="record layout resident in HBM (identical results): i32 = 8 x int32 = 32 B per record; u24 = 8 x 24 bits = "
                         "24 B (counts <= 2^24 - 2, i.e. everything the fast kernels accept); u16 = 8 x uint16 = 16 B (counts <= 65534). "
                         "auto (default) = u24 when the cohort fits, else i32.  The cohort is packed once at setup "
                         "(ampli_records_pack24/16); at N = 1 the other layouts are timed too, outside the timed region")
    ap.add_argument("--merge", default="sliced", choices=["sliced", "allreduce"],
                    help="N>1 exchange: sliced = reduce-scatter + all-to-all + all-gather by position slices (default); "
                         "allreduce = one packed all-reduce + all-gather of whole germ-max regions")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N>1 on one GPU)")
    ap.add_argument("--group", type=int, default=0, help="N>1, sliced merge: independent batches per round of collectives "
                    "(fewer, larger RCCL messages and fewer cross-stream waits per batch); 0 = auto: 4 for runs of >= 16 steps, "
                    "2 for >= 8, else 1 (the last group's exchange has nothing to hide behind, so short runs want small groups)")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: run the N>1 code path (process group, merge, pipelined loop) "
                    "even with one rank -- over RCCL this exercises the real collectives on a one-GPU box")
    ap.add_argument("--check", action="store_true", help="N>1: verify the merged table against a single-pass reduction of all shards")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------
# CPU baseline leg (rank 0, N=1): the reference's own error-estimation code when oracle/_ref holds it,
# and the oracle port for both halves.  Bounded sample of the same synthetic workload.
# ----------------------------------------------------------------------------------------------------
def _write_aseq_dir(recs, chrom_pos, d):
    """recs [S][P][8] numpy -> S .PILEUP.ASEQ files (columns as EE:1149)."""
    import numpy as np

    os.makedirs(d, exist_ok=True)
    S, P, _ = recs.shape
    chroms = np.array([c for c, _ in chrom_pos])
    poss = np.array([p for _, p in chrom_pos])
    for s in range(S):
        r = recs[s]
        present = r[:, 0] != np.iinfo(np.int32).min
        fw, bw = r[:, :4].astype(np.int64), r[:, 4:].astype(np.int64)
        tot = fw + bw
        rd = tot.sum(1)
        with open(os.path.join(d, f"N{s:04d}.PILEUP.ASEQ"), "w") as f:
            f.write("chr\tpos\tdbsnp\tMAF\tref\talt\tA\tC\tG\tT\tRD\tArs\tCrs\tGrs\tTrs\n")
            lines = [f"{chroms[p]}\t{poss[p]}\t.\t.\t.\t.\t{tot[p,0]}\t{tot[p,1]}\t{tot[p,2]}\t{tot[p,3]}\t{rd[p]}\t"
                     f"{bw[p,0]}\t{bw[p,1]}\t{bw[p,2]}\t{bw[p,3]}\n" for p in range(P) if present[p]]
            f.write("".join(lines))


def synthetic_panel(P, n_regions=30):
    """30 amplicon regions on chr1..22,X cycling, ceil(P/30) consecutive 1-based positions each."""
    chroms = [f"chr{i}" for i in range(1, 23)] + ["chrX"]
    per = -(-P // n_regions)
    rows, pos = [], []
    for i in range(n_regions):
        n = min(per, P - i * per)
        if n <= 0:
            break
        c = chroms[i % len(chroms)]
        start = 1_000_000 + (i // len(chroms)) * 5_000_000
        rows.append((c, start, start + n - 1))
        pos += [(c, start + j) for j in range(n)]
    return rows, pos


def cpu_baseline(cfg):
    import numpy as np

    from oracle import pyoracle as orc
    from tests.helpers import synth_recs, synth_ref

    out = {}
    # --- port (oracle) on both halves -----------------------------------------------------------
    P, S, T = 50_000, 128, 48
    normals = synth_recs(P, S, seed=SEED, depth=cfg["depth"])
    tumours = synth_recs(P, T, seed=SEED, depth=cfg["depth"], tumour=True)
    ref = synth_ref(P, seed=SEED)
    t0 = time.perf_counter()
    acc = orc.error_reduce(normals, P, 0.002, 100)
    fin = orc.error_finalize(acc)
    t1 = time.perf_counter()
    orc.poisson_call(tumours, P, fin["thr"], ref, 100, dense=False)
    t2 = time.perf_counter()
    port = dict(value=(P * S + P * T) / (t2 - t0), unit="position-evaluations/s", cores=1, kind="port",
                sample=f"oracle/ampli_oracle.c on {P} positions x {S} normals + {T} tumours of the same synthetic panel "
                       f"(error-est {t1 - t0:.3f} s, calling {t2 - t1:.3f} s)")
    out = port
    # --- the reference's own code for the error-estimation half --------------------------------
    drv = orc.REF_EE_DRIVER
    if os.path.exists(drv):
        Pr, Sr = 20_000, 64
        rows, pos = synthetic_panel(Pr)
        recs = synth_recs(Pr, Sr, seed=SEED, depth=cfg["depth"])
        refb = synth_ref(Pr, seed=SEED)
        with tempfile.TemporaryDirectory(prefix="ampli_ref_") as d:
            _write_aseq_dir(recs, pos, os.path.join(d, "normals"))
            with open(os.path.join(d, "panel.bed"), "w") as f:
                f.write("".join(f"{c}\t{a}\t{b}\tAMPL{i}\trs{i}\tGENE{i}\n" for i, (c, a, b) in enumerate(rows)))
            with open(os.path.join(d, "refbases.txt"), "w") as f:
                f.write("".join(f"{c}\t{p}\t{'ACGT'[refb[i]]}\n" for i, (c, p) in enumerate(pos)))
            open(os.path.join(d, "dups.txt"), "w").close()
            os.makedirs(os.path.join(d, "out"))
            r = subprocess.run([drv, os.path.join(d, "panel.bed"), os.path.join(d, "refbases.txt"), os.path.join(d, "dups.txt"),
                                os.path.join(d, "normals"), "0.002", "100", os.path.join(d, "out")],
                               capture_output=True, text=True, cwd=d)
            tm = {ln.split()[1]: float(ln.split()[2]) for ln in r.stderr.splitlines() if ln.startswith("TIMING")}
        if r.returncode == 0 and "storeGermlineStatistics" in tm:
            t_ref = tm["storeGermlineStatistics"] + tm["estimateThresholds"] + tm["generateFinalOutput"]
            out = dict(value=tm["records"] / t_ref, unit="position-evaluations/s", cores=1, kind="reference",
                       sample=f"reference AmpliSolveErrorEstimation.cpp compiled -O2 from /root/reference (oracle/_ref/ee_ref_driver: "
                              f"storeGermlineStatistics+estimateThresholds+generateFinalOutput, samtools step excluded) on "
                              f"{Pr} positions x {Sr} normals of the same synthetic panel = {int(tm['records'])} records in {t_ref:.2f} s; "
                              f"error-estimation half only: the calling half of the reference needs Boost (unbuildable here), "
                              f"see 'port' for the oracle on both halves",
                       port=port)
    return out


# ----------------------------------------------------------------------------------------------------
def main():
    args = parse_args()
    cfg = CONFIGS[args.config]
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    multi = world > 1 or args.force_dist  # the N>1 code path (also runs with one rank under --force-dist)
    import torch
    import torch.distributed as dist

    from amplisolve_amd import Context
    from amplisolve_amd.api import POISSON_FULL, POISSON_PREFILTER
    from amplisolve_amd.dist import merge_error_table

    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(ndev, 1)  # == local_rank on a full node; ranks share a device only in rehearsals
    torch.cuda.set_device(dev_index)
    if multi:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend)
    # everything (kernels, torch plumbing, RCCL's stream dependencies) hangs off ONE non-default stream: the legacy
    # null stream carries implicit synchronisation that costs a few us per launch
    main_stream = torch.cuda.Stream(device=dev_index)
    torch.cuda.set_stream(main_stream)
    ctx = Context(dev_index)
    if args.splits or args.groups:
        ctx.set_tuning(args.splits, groups=args.groups)
    mode = POISSON_PREFILTER if args.mode == "prefilter" else POISSON_FULL

    P, S, T, depth = cfg["P"], cfg["S"], cfg["T"], cfg["depth"]
    # synthetic shard of this rank, generated in HBM (bit-identical to the host generator)
    normals = ctx.synth_fill(P, S, first_sample=rank * S, seed=SEED, depth=depth)
    tumours = ctx.synth_fill(P, T, first_sample=rank * T, seed=SEED, depth=depth, tumour=True)
    ref_code = ctx.synth_ref(P, seed=SEED)
    # record layout: a resident cohort is packed ONCE (data preparation, like the H2D copy it replaces) and then
    # evaluated many times; identical results in every layout (tests/test_gpu_u16.py)
    REC_BYTES = {"i32": 32, "u24": 24, "u16": 16}
    packed = {"i32": (normals, tumours)}
    for name in ("u24", "u16"):
        pn, ok_n = ctx.pack(normals, name)
        pt, ok_t = ctx.pack(tumours, name)
        fits = torch.tensor([1 if (ok_n and ok_t) else 0], dtype=torch.int32, device=ctx.device)
        if multi:
            dist.all_reduce(fits, op=dist.ReduceOp.MIN)  # one layout for the whole job
        if int(fits.item()):
            packed[name] = (pn, pt)
        del pn, pt
    layout = args.records if args.records != "auto" else ("u24" if "u24" in packed else "i32")
    if layout not in packed:
        raise SystemExit(f"--records {layout}: a count of this workload does not fit that layout")
    normals, tumours = packed[layout]
    ctx.set_record_layout(layout)
    if multi or args.streams > 1:
        packed = {layout: packed[layout]}  # the other layouts are only kept for the N = 1 comparison legs
    rec_bytes = REC_BYTES[layout]
    accs = [ctx.new_acc(P) for _ in range(2 if multi else 1)]
    for a in accs:
        a.buf.zero_()
    acc = accs[0]
    fin = None
    call_mask = torch.empty((T, P), dtype=torch.uint8, device=ctx.device)
    cap = 1 << 20
    from amplisolve_amd._lib import Call
    from amplisolve_amd.api import CALL_COUNTER_STRIDE, CALL_COUNTER_WORDS
    from amplisolve_amd.dist import TableMerger
    import ctypes

    calls_buf = torch.empty((cap * ctypes.sizeof(Call),), dtype=torch.uint8, device=ctx.device)
    n_calls = torch.zeros((CALL_COUNTER_WORDS,), dtype=torch.int64, device=ctx.device)
    from amplisolve_amd.dist import SlicedMerger

    sliced = multi and args.merge == "sliced"
    G = args.group if args.group > 0 else (4 if args.steps >= 16 else 2 if args.steps >= 8 else 1)  # batches per round of collectives
    merger = None
    if multi:
        merger = (SlicedMerger(P, world, rank, ctx.device, batches=G) if sliced else
                  TableMerger(P, world, ctx.device, ctx.gm_merge, pack=ctx.acc_pack, unpack=ctx.acc_unpack))

    ev = [[ctx.event() for _ in range(4)] for _ in range(args.steps)]
    ev_steps = [i for i in range(args.steps) if i % max(1, args.event_every) == 0]

    # two error tables, used alternately: with the asynchronous drain the survivors of batch i are still being scored
    # (reading batch i's thresholds) while batch i+1's table is being written
    fins = [None, None]
    if args.async_drain:
        ctx.set_async_drain(True)

    last_blocks = [None]

    def reduce_part(i, timed, slot):
        nonlocal fin
        timed = timed and i % max(1, args.event_every) == 0
        if timed:
            ctx.record(ev[i][0])
        if not multi:  # the panel lives on one device: finalize fused into the reduce epilogue (ampli_error_estimate)
            fins[i & 1] = fin = ctx.error_estimate(normals, P, 0.002, 100, out=fins[i & 1])
        elif sliced:  # shard of a multi-GPU panel: sums and germ-max pairs straight into the slice-major exchange buffers
            ctx.error_reduce_sliced(normals, P, world, merger.sums[slot], merger.gm[slot], 0.002, 100, first_sample=rank * S)
        else:  # shard of a multi-GPU panel: sums straight into the all-reduce buffer, gm planes into the table
            ctx.error_reduce_packed(normals, P, accs[slot], merger.packed[slot], 0.002, 100, first_sample=rank * S)
        if timed:
            ctx.record(ev[i][1])

    def call_part(i, timed, slot):
        nonlocal fin
        timed = timed and i % max(1, args.event_every) == 0
        if sliced:
            # poisson_call reads the thresholds straight from the gathered blocks (the blocks ARE the error table, by
            # position slice); the plane-major form is only materialised where somebody wants it (--check, the flags)
            if timed:
                ctx.record(ev[i][2])
            ctx.poisson_call(tumours, P, merger.blocks[slot], ref_code, 100, mode=mode, call_mask=call_mask, capacity=cap,
                             calls_buf=calls_buf, n_calls=n_calls, blocks_of=world)
            if timed:
                ctx.record(ev[i][3])
            last_blocks[0] = (merger.blocks[slot], i % G)
            return
        elif multi:  # finalize straight from the all-reduced sums + gathered germ-max regions
            fins[i & 1] = fin = ctx.error_finalize_merged(P, merger.packed[slot], merger.gathered[slot], world, 0.002, 100, out=fins[i & 1])
        if timed:
            ctx.record(ev[i][2])
        ctx.poisson_call(tumours, P, fin.thr, ref_code, 100, mode=mode, call_mask=call_mask, capacity=cap,
                         calls_buf=calls_buf, n_calls=n_calls)
        if timed:
            ctx.record(ev[i][3])

    def run_steps(n, timed):
        """n passes of the hot path.  N == 1: strictly sequential.  N > 1: software-pipelined across the independent
        batches -- the table merge of batch i (RCCL, own stream) overlaps finalize + poisson_call of batch i-1 and
        error_reduce of batch i+1; every batch still goes through every stage inside the timed region."""
        if n <= 0:
            return
        if not multi:
            for i in range(n):
                reduce_part(i, timed, 0)
                call_part(i, timed, 0)
            return
        if sliced:
            # three GROUPS of G batches in flight: G x reduce -> [reduce-scatter + all-to-all](group) | G x finalize_slice ->
            # [all-gather](group - 1) | G x poisson_call(group - 2); one round of collectives serves the G batches of a group,
            # and each collective has G whole error_reduce launches between its start and its wait
            hx, hg = {}, {}
            ngroups = (n + G - 1) // G

            def members(gi):
                return range(gi * G, min(n, (gi + 1) * G))

            def mid(gi):
                sj = gi % 3
                merger.wait(hx.pop(gi))
                for i in members(gi):
                    ctx.set_slice_group(G, i % G)
                    ctx.error_finalize_slice(P, world, rank, merger.sum_slice[sj], merger.gm_recv[sj], merger.block[sj], 0.002, 100)
                hg[gi] = merger.start_gather(sj)

            def last(gi):
                merger.wait(hg.pop(gi))
                for i in members(gi):
                    ctx.set_slice_group(G, i % G)
                    call_part(i, timed, gi % 3)

            # issue order on RCCL's (in-order) stream: all-gather(group - 1) BEFORE the exchange of this group, so that the
            # poisson_calls of group - 1 do not wait behind this group's reduce-scatter -- in the steady state and,
            # above all, at the end of the run, where the last exchange then hides behind the previous group's calls
            for gi in range(ngroups):
                for i in members(gi):
                    ctx.set_slice_group(G, i % G)
                    reduce_part(i, timed, gi % 3)
                if gi >= 1:
                    mid(gi - 1)
                hx[gi] = merger.start_exchange(gi % 3)
                if gi >= 2:
                    last(gi - 2)
            if ngroups >= 2:
                last(ngroups - 2)
            mid(ngroups - 1)
            last(ngroups - 1)
            return
        pending = None
        for i in range(n):
            slot = i & 1
            reduce_part(i, timed, slot)
            h = merger.start(accs[slot], slot, prepacked=True)
            if pending is not None:
                j, pslot, ph = pending
                merger.wait(ph)
                call_part(j, timed, pslot)
            pending = (i, slot, h)
        j, pslot, ph = pending
        merger.wait(ph)
        call_part(j, timed, pslot)

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    lanes = None
    if not multi and args.streams > 1:
        # extra lanes: own stream + own outputs each; inputs are shared (read-only)
        lanes = []
        for _ in range(args.streams):
            st = torch.cuda.Stream(device=dev_index)
            with torch.cuda.stream(st):
                c = Context(dev_index)
                c.set_record_layout(layout)
                f = c.error_estimate(normals, P, 0.002, 100)
                r = c.poisson_call(tumours, P, f.thr, ref_code, 100, mode=mode, capacity=cap)
            lanes.append((c, f, r))
        torch.cuda.synchronize()

        def run_steps(n, timed):  # noqa: F811  (replaces the single-stream loop)
            for i in range(n):
                c, f, r = lanes[i % len(lanes)]
                t = timed and i % max(1, args.event_every) == 0
                if t:
                    c.record(ev[i][0])
                c.error_estimate(normals, P, 0.002, 100, out=f)
                if t:
                    c.record(ev[i][1])
                    c.record(ev[i][2])
                c.poisson_call(tumours, P, f.thr, ref_code, 100, mode=mode, call_mask=r["call_mask"], capacity=r["capacity"],
                               calls_buf=r["calls_buf"], n_calls=r["n_calls"])
                if t:
                    c.record(ev[i][3])

    try:
        run_steps(args.warmup, False)
        fence()
    except Exception as exc:  # noqa: BLE001
        # a runtime that rejects the sliced exchange's collectives does so on every rank at the first call:
        # fall back, loudly, to the all-reduce form rather than lose the N > 1 measurement
        if not sliced:
            raise
        print(f"rank {rank}: sliced merge failed in warm-up ({type(exc).__name__}: {exc}); falling back to --merge allreduce", file=sys.stderr)
        sliced = False
        args.merge = "allreduce"
        merger = TableMerger(P, world, ctx.device, ctx.gm_merge, pack=ctx.acc_pack, unpack=ctx.acc_unpack)
        run_steps(args.warmup, False)
        fence()
    def materialise():
        """sliced merge: the plane-major table of the last finished batch (outside the per-batch work)"""
        nonlocal fin
        if sliced and last_blocks[0] is not None:
            blocks, g = last_blocks[0]
            ctx.set_slice_group(G, g)
            fin = ctx.error_table_unslice(P, world, blocks)
            ctx.set_slice_group(1, 0)
            fins[(args.warmup - 1) & 1] = fin

    materialise()
    if multi and args.check:
        # every shard regenerated locally and reduced in one pass must equal the merged table, bit for bit
        allrecs = torch.cat([ctx.synth_fill(P, S, first_sample=k * S, seed=SEED, depth=depth) for k in range(world)])
        allrecs, _ = ctx.pack(allrecs, layout)
        ref = ctx.error_estimate(allrecs, P, 0.002, 100)
        got = fins[(args.warmup - 1) & 1]
        present = ref.germ_present > 0
        for name, ok in (("rate", torch.equal(ref.rate, got.rate)), ("code", torch.equal(ref.code, got.code)),
                         ("thr", torch.equal(ref.thr, got.thr)), ("germ_present", torch.equal(ref.germ_present, got.germ_present)),
                         ("germ_val", torch.equal(ref.germ_val[present], got.germ_val[present]))):
            if not ok:
                raise SystemExit(f"rank {rank}: merged error table differs from the single-pass one in {name}")
        del allrecs, ref
        if rank == 0:
            print("check: merged error table == single-pass error table (rate, code, thr, germ-max), bit for bit", file=sys.stderr)
        fence()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    fence()
    elapsed = time.perf_counter() - t0
    if multi:
        te = torch.tensor([elapsed], dtype=torch.float64, device=ctx.device)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    materialise()
    t_red = sum(ctx.elapsed_ms(ev[i][0], ev[i][1]) for i in ev_steps) / len(ev_steps)
    t_call_main = sum(ctx.elapsed_ms(ev[i][2], ev[i][3]) for i in ev_steps) / len(ev_steps)  # main-stream part (all of it unless --async-drain)
    ctx.wait_calls()
    if lanes is not None:
        n_calls, fin = lanes[0][2]["n_calls"], lanes[0][1]
    n_found = int(n_calls[::CALL_COUNTER_STRIDE].sum().item())
    # the whole poisson_call (stream + drain kernels back to back), outside the timed region
    ctx.set_async_drain(False)
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0)
    for _ in range(5):
        ctx.poisson_call(tumours, P, fin.thr, ref_code, 100, mode=mode, call_mask=call_mask, capacity=cap, calls_buf=calls_buf, n_calls=n_calls)
    ctx.record(e1)
    t_call_sep = ctx.elapsed_ms(e0, e1) / 5
    # without --async-drain the whole call sits on the main stream and the in-region HIP events are the measurement
    t_call = t_call_sep if args.async_drain else t_call_main
    # validation mode, outside the timed region: all six scores of every record, as the reference evaluates them
    e0, e1 = ctx.event(), ctx.event()
    n_calls.zero_()
    ctx.record(e0)
    for _ in range(3):
        ctx.poisson_call(tumours, P, fin.thr, ref_code, 100, mode=POISSON_FULL, call_mask=call_mask)
    ctx.record(e1)
    t_call_full = ctx.elapsed_ms(e0, e1) / 3
    flags = int(fin.flags.item())
    if flags != 0:
        raise SystemExit("error_finalize reported an exactness-envelope violation")

    others = []
    for name in [n for n in ("i32", "u24", "u16") if n in packed and n != layout and not multi and lanes is None]:
        # the same workload in the other record layouts, outside the timed region, for comparison
        an, at = packed[name]
        c2 = Context(dev_index)
        c2.set_record_layout(name)
        f2 = c2.error_estimate(an, P, 0.002, 100)
        r2 = c2.poisson_call(at, P, f2.thr, ref_code, 100, mode=mode, capacity=cap)
        same = all(torch.equal(getattr(f2, k).view(torch.uint8), getattr(fin, k).view(torch.uint8)) for k in ("rate", "thr", "code", "germ_present"))
        same = same and torch.equal(r2["call_mask"], call_mask)

        def avg_ms(fn, reps):
            fn()
            a, b = c2.event(), c2.event()
            c2.record(a)
            for _ in range(reps):
                fn()
            c2.record(b)
            return c2.elapsed_ms(a, b) / reps

        def step2():
            c2.error_estimate(an, P, 0.002, 100, out=f2)
            c2.poisson_call(at, P, f2.thr, ref_code, 100, mode=mode, call_mask=r2["call_mask"], capacity=cap,
                            calls_buf=r2["calls_buf"], n_calls=r2["n_calls"])

        ob = REC_BYTES[name]
        t2_red = avg_ms(lambda: c2.error_estimate(an, P, 0.002, 100, out=f2), 20)
        t2_step = avg_ms(step2, 20)
        others.append({"records": f"{name} ({ob} B per record)", "ms_per_step": t2_step, "value": (P * S + P * T) / (t2_step * 1e-3),
                       "error_reduce_ms": t2_red, "error_reduce_GBs": (ob * P * S + 88 * P) / (t2_red * 1e-3) / 1e9,
                       "error_reduce_frac_of_peak": (ob * P * S + 88 * P) / (t2_red * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "same_outputs": bool(same), "note": "20 passes, HIP events, outside the timed region"})
        c2.close()
        del an, at, f2, r2

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        evals = world * (P * S + P * T) * args.steps
        acc_bytes = ctx.lib.ampli_acc_bytes(P)
        # algorithmic bytes of error_reduce per launch: the records + what it writes (the accumulator table,
        # or at N = 1 the finalised error table: rate 32 B + thr 32 B + code 4 B + germ 16+4 B per position)
        # (N > 1, sliced merge: 21 doubles + 8 floats per position into the exchange buffers = 200 B)
        red_bytes = rec_bytes * P * S + ((200 * P if sliced else acc_bytes) if multi else 88 * P)
        call_bytes = rec_bytes * P * T + 33 * P + P * T    # poisson_call: records + thresholds/ref + mask
        lay = {"i32": 0, "u16": 1, "u24": 2}[layout]
        if t_red >= t_call:
            dom, dom_ms, dom_bytes = f"error_reduce_kernel<true, 1, {lay}>", t_red, red_bytes
        else:
            dom, dom_ms, dom_bytes = f"poisson_stream_kernel<{lay}>+poisson_drain_kernel", t_call, call_bytes
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        # HBM traffic of the dominant kernel from the PMC counters: collected with rocprofv3 --pmc in separate passes of
        # this same command (tools/collect_profiles.sh) and corrected as MI355X_MICROARCH.md prescribes; committed
        # under profiles/.  null when the workload is not the profiled one.
        traffic = None
        pj = os.path.join(ROOT, "profiles", "r01", "pmc_summary.json")
        if args.config == "c3" and not multi and os.path.exists(pj):
            try:
                pm = json.load(open(pj))
                traffic = sum(pm[k].get("hbm_bytes_per_launch", 0.0) for k in pm
                              if any(part in k for part in dom.split("+"))) or None
            except Exception:
                traffic = None
        out = {
            "metric": "position-evaluations/s (error-est + Poisson call)",
            "value": evals / elapsed,
            "unit": "position-evaluations/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"i32": "int32", "u24": "24-bit", "u16": "uint16"}[layout] + " counts; f64 sums / Poisson; f32 rates",
            "data": "synthetic",
            "config": {"workload": cfg["name"], "positions": P, "normals_per_gpu": S, "tumours_per_gpu": T, "depth": depth,
                       "C_value": 0.002, "coverage_cutoff": 100, "poisson_mode": args.mode, "streams": args.streams if not multi else 1,
                       "parallelism": f"tumour+normal sample shards x{world}" + (("; per batch: RCCL reduce-scatter of the sums + all-to-all of the germ-max pairs by position slice, finalize of the own slice, all-gather of the error table -- one round of collectives per group of independent batches, three groups in flight" if sliced else "; one packed RCCL all-reduce + all-gather of the germ-max regions per batch, overlapped with the neighbouring batches") if multi else ""),
                       "merge": (args.merge if multi else None), "batches_per_exchange": (G if sliced else None),
                       "rehearsal": ("N>1 code path forced on one rank (--force-dist)" if args.force_dist and world == 1 else None),
                       "records": f"{layout} ({rec_bytes} B per record: 8 fields x {rec_bytes // 8 * 8} bits)"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "avg_ms": dom_ms, "algorithmic_bytes": dom_bytes,
                         "traffic_source": "profiles/r01/pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 x2 read correction)" if traffic else None},
            "kernels": {"error_reduce_ms": t_red, "error_reduce_GBs": red_bytes / (t_red * 1e-3) / 1e9,
                        "poisson_call_ms": t_call, "poisson_call_GBs": call_bytes / (t_call * 1e-3) / 1e9,
                        "poisson_call_main_stream_ms": t_call_main, "drain": "side stream, overlapped with the next batch" if args.async_drain else "main stream",
                        "poisson_call_full_mode_ms": t_call_full,
                        # SURVEY.md 8d's three rates: panel positions/s through error estimation, tumour position-evaluations/s
                        # through calling, and tumour position-evaluations/s through the whole pass
                        "R_EE_positions_per_s": P / (t_red * 1e-3), "R_VC_evals_per_s": P * T / (t_call * 1e-3),
                        "R_pipe_evals_per_s": world * P * T / (ms_per_step * 1e-3)},
            "calls_per_step": n_found,
        }
        if others:
            out["other_record_layouts"] = others
        if not multi and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(cfg)
            except Exception as e:  # the baseline leg must never take the GPU number down with it
                out["cpu_baseline"] = {"value": None, "unit": "position-evaluations/s", "cores": 1, "kind": "port",
                                       "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
