"""error_reduce / poisson_call rate against the panel size P, with the dense row stride (P records) and with a padded one
(P + 64 records, ampli_records.row_stride): when a sample row is a large power of two long (P = 262 144 x 32 B = 8 MiB) the four
waves of a workgroup, which read the same positions of four different rows, collide on HBM channels; 64 records of padding
per row (what the host packer adds for such panels) restore the rate.   usage: python tools/sweep_P.py [i32|u24|u16 ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

from amplisolve_amd import Context

S, T = 256, 96
RB = {"i32": 32, "u24": 24, "u16": 16}
base = Context(0)
for layout in (sys.argv[1:] or ["i32", "u24"]):
    ctx = Context(0)
    ctx.set_record_layout(layout)
    for P in (98304, 100000, 131072, 262144, 524288):
        nor32 = base.synth_fill(P, S)
        ref = base.synth_ref(P)
        for pad in (0, 64):
            stride = P + pad
            rows = nor32.view(S, P, 8)
            if pad:
                padded = torch.empty((S, stride, 8), dtype=torch.int32, device="cuda")
                padded[:, :P] = rows
                padded[:, P:] = rows[:, :pad]
                rows = padded
            packed, _ = ctx.pack(rows.contiguous().view(S, stride, 8), layout)
            rec = ctx.records(packed, layout, S, row_stride=stride)
            out = ctx.error_reduce_records(rec, P, None, finalize=True)
            for _ in range(3):
                ctx.error_reduce_records(rec, P, None, out=out, finalize=True)
            e0, e1 = ctx.event(), ctx.event()
            ctx.record(e0)
            for _ in range(20):
                ctx.error_reduce_records(rec, P, None, out=out, finalize=True)
            ctx.record(e1)
            torch.cuda.synchronize()
            ms = ctx.elapsed_ms(e0, e1) / 20
            nbytes = RB[layout] * P * S + 88 * P
            print(f"{layout} P={P:7d} row stride {stride:7d} records ({stride * RB[layout] / 2**20:6.2f} MiB): error_estimate {ms * 1e3:7.1f} us = "
                  f"{nbytes / ms / 1e9:5.2f} TB/s", flush=True)
            del packed, rec
        del nor32
    ctx.close()
