"""error_estimate time against the number of 64-position tiles (uint16 records, S = 256): how much of config 3's launch is the
partly filled last round of workgroups.  A CU holds 5 workgroups of the compact kernel (96 VGPRs) and 4 of the general one
(120), so the chip holds 1280 / 1024 at once and config 3 (1563 tiles) is 1.22 / 1.53 rounds.
usage: python tools/sweep_tiles.py [S]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

from amplisolve_amd import Context

S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = Context(0)
ctx.set_record_layout("u16")
TILES = (64, 128, 283, 512, 768, 1024, 1280, 1536, 1563, 1792, 2048, 2560, 3840, 5120)
for tiles in TILES:
    P = tiles * 64 if tiles != 1563 else 100000
    nor32 = ctx.synth_fill(P, S)
    packed, _ = ctx.pack(nor32.view(S, P, 8), "u16")
    rec = ctx.records(packed, "u16", S)
    row = []
    # compact kernel with one sample split forced | the launch the library picks by itself | general kernel, library's pick
    for compact, splits, groups in ((1, 1, 1), (1, 0, 0), (0, 0, 0)):
        ctx.set_reduce_compact(compact)
        ctx.set_tuning(splits, False, groups)
        out = ctx.error_reduce_records(rec, P, None, finalize=True)
        for _ in range(3):
            ctx.error_reduce_records(rec, P, None, out=out, finalize=True)
        best = 1e9
        for _ in range(3):
            e0, e1 = ctx.event(), ctx.event()
            ctx.record(e0)
            for _ in range(20):
                ctx.error_reduce_records(rec, P, None, out=out, finalize=True)
            ctx.record(e1)
            torch.cuda.synchronize()
            best = min(best, ctx.elapsed_ms(e0, e1) / 20)
        row.append(best * 1e3)
    nbytes = 16 * P * S + 88 * P
    print(f"tiles {tiles:5d} (P = {P:7d}; {tiles / 1280:4.2f} / {tiles / 1024:4.2f} rounds): compact, one split {row[0]:6.1f} us = {nbytes / row[0] / 1e6:5.2f} TB/s"
          f" ({row[0] * 1e3 / tiles:5.1f} ns per tile)   library's choice {row[1]:6.1f} us   general kernel {row[2]:6.1f} us = {nbytes / row[2] / 1e6:5.2f} TB/s", flush=True)
    del nor32, packed, rec
ctx.close()
