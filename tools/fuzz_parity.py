"""Randomised parity sweep on the GPU: random panel shapes, parameters, kernel variants, record layouts, shard counts
and thresholds, every result compared with the CPU oracle (bit-exact planes / masks, 1e-6 on Q).  Test infrastructure:
    python tools/fuzz_parity.py --seconds 300 --seed 1
Prints one line per case family and a summary; exits non-zero on the first mismatch (with the case's seed)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from amplisolve_amd import Context
from amplisolve_amd.api import POISSON_FULL, POISSON_PREFILTER
from amplisolve_amd.dist import shard_range, slice_geometry
from oracle import pyoracle as orc
from tests.helpers import edge_case_recs, synth_recs


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x)).cuda()


def eq_bits(a, b):
    return np.array_equal(np.ascontiguousarray(a).view(np.uint8), np.ascontiguousarray(b).view(np.uint8))


def check_final(fin, ref, what):
    ok = eq_bits(fin.code.cpu().numpy(), ref["code"]) and eq_bits(fin.rate.cpu().numpy(), ref["rate"]) and eq_bits(fin.thr.cpu().numpy(), ref["thr"])
    ok = ok and eq_bits(fin.germ_present.cpu().numpy(), ref["germ_present"])
    m = ref["germ_present"] > 0
    ok = ok and np.array_equal(fin.germ_val.cpu().numpy()[m].astype(np.float64), ref["germ_val"][m])
    if not ok:
        raise AssertionError(what)


def make_records(rng, P, E, S, kind, u16):
    R = P + E
    if kind == "synth":
        recs = synth_recs(R, S, seed=int(rng.integers(1, 1 << 30)), depth=int(rng.choice([300, 2000, 9000])))
    else:
        recs = edge_case_recs(R, S, rng)
    if rng.random() < 0.3:  # sprinkle big depths (still inside the fast kernel's envelope, and inside uint16 when asked)
        k = int(rng.integers(1, 20))
        hi = 60000 if u16 else 1_500_000
        for s_i, r_i in zip(rng.integers(0, S, k), rng.integers(0, R, k)):
            a = int(rng.integers(100, hi // 20))
            recs[s_i, r_i] = [hi, a, int(rng.integers(0, 5)), 0, hi - 7, a + 3, 0, int(rng.integers(0, 9))]
    return recs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--only", type=int, default=None, help="replay one case by its printed seed")
    ap.add_argument("--big", action="store_true", help="larger panels and cohorts (fewer cases per second)")
    args = ap.parse_args()
    ctxs = {"i32": Context(0), "u16": Context(0), "u24": Context(0)}
    for k, c in ctxs.items():
        c.set_record_layout(k)
    t_end = time.time() + args.seconds
    n = {"reduce": 0, "sliced": 0, "poisson": 0}
    case = 0
    while time.time() < t_end:
        seed = args.seed * 1_000_003 + case if args.only is None else args.only
        case += 1
        if args.only is not None and case > 1:
            break
        rng = np.random.default_rng(seed)
        try:
            if args.big:
                P = int(rng.choice([4097, 20_000, 65_536, int(rng.integers(1000, 60_000))]))
                S = int(rng.choice([5, 33, 64, 130, 300, 700, 1100]))
                cap_cells = 6_000_000
            else:
                P = int(rng.choice([1, 7, 63, 64, 65, 255, 256, 257, 1000, 4097, int(rng.integers(1, 6000))]))
                S = int(rng.choice([1, 2, 3, 4, 5, 8, 17, 33, 64, 130, 300]))
                cap_cells = 600_000
            if P * S > cap_cells:
                S = max(1, cap_cells // P)
            dups = rng.random() < 0.4
            mult = np.zeros(P, np.int64)
            if dups:
                mult[rng.random(P) < 0.05] = 1
                mult[rng.random(P) < 0.01] = 2
            dup_off = np.concatenate([[0], np.cumsum(mult)]).astype(np.uint32)
            E = int(dup_off[-1])
            layout = str(rng.choice(["i32", "u16", "u24"]))
            u16 = layout == "u16"
            kind = "synth" if rng.random() < 0.5 else "edge"
            recs = make_records(rng, P, E, S, kind, u16)
            C_value = float(rng.choice([0.002, 0.01, 0.0005, 0.05]))
            cov = int(rng.choice([1, 50, 100, 500, 1000]))
            ctx = ctxs[layout]
            dev, fits = ctx.pack(t(recs), layout)
            if not fits:
                raise AssertionError("records do not fit the layout although generated to")
            d_dup = t(dup_off) if E else None
            o_acc = orc.error_reduce(recs, P, C_value, cov, E=E, dup_off=dup_off if E else None)
            o_fin = orc.error_finalize(o_acc)
            def check_inorder():
                """round 6: the eight threshold sums in the reference's own order of addition (ampli_error_sums_inorder), the cohort cut into
                random chunks walked from the last to the first -- bit for bit the oracle's, inside the exactness envelope and outside it"""
                want = orc.error_sums_inorder(recs, P, C_value, cov, E=E, dup_off=dup_off if E else None)
                ncut = int(rng.integers(1, min(S, 4) + 1))
                cuts = [0] + sorted(rng.choice(np.arange(1, S), size=ncut - 1, replace=False).tolist()) + [S] if ncut > 1 else [0, S]
                acc_io = ctx.new_acc(P)
                acc_io.buf.fill_(0x33)
                for k in range(len(cuts) - 2, -1, -1):
                    rec_k = ctx.records(dev[cuts[k]:cuts[k + 1]], layout, cuts[k + 1] - cuts[k], E=E, row_stride=P + E, dup_off=d_dup)
                    ctx.error_sums_inorder(rec_k, P, acc_io, C_value, cov, accumulate=k != len(cuts) - 2)
                if not eq_bits(acc_io.snt.cpu().numpy(), want):
                    raise AssertionError("in-order threshold sums differ from the oracle's")
                n["inorder"] = n.get("inorder", 0) + 1

            if o_acc["order_sensitive"]:
                check_inorder()  # outside the exactness envelope by construction (tiny C, cov 1): only the reference's own order is defined
                n["inorder_outside_envelope"] = n.get("inorder_outside_envelope", 0) + 1
                continue
            if rng.random() < 0.15:
                check_inorder()
            # ---- reduce variants ----
            general = bool(rng.random() < 0.3)
            groups = int(rng.choice([0, 1, 2, 4]))
            splits = int(rng.choice([0, 0, 1, 2, 3]))
            if layout != "i32" and rng.random() < 0.5:
                general, groups, splits = False, 1, 1  # the shape error_reduce_u16_kernel takes (the fused estimate below); with E > 0 its tiles
                                                       # of positions listed more than once go to the general kernel over the tile list
            ctx.set_tuning(splits, general=general, groups=groups)
            acc = ctx.error_reduce(dev, P, C_value, cov, E=E, dup_off=d_dup)
            fin2 = ctx.error_finalize(acc, C_value, cov)
            fin1 = ctx.error_estimate(dev, P, C_value, cov, E=E, dup_off=d_dup)
            n["compact_kernel"] = n.get("compact_kernel", 0) + (ctx.last_reduce_kernel() in ("error_reduce_u16_kernel", "error_reduce_u24_kernel"))
            ctx.set_tuning(0)
            flags = ctx.flags()
            if flags & 2:
                raise AssertionError("fast kernel asked for a rerun on records inside its envelope")
            if (int(fin1.flags.item()) | int(fin2.flags.item())) & 1:
                # the double sums left the exactness envelope (tiny C x cutoff 1 with deep records): the product says so
                # (flag bit 0; the command line refuses to write the table) and equality with the oracle is not promised
                n["envelope_flagged"] = n.get("envelope_flagged", 0) + 1
                continue
            for name in ("snt", "srd", "cnt", "nrec", "gm_n"):
                if not eq_bits(getattr(acc, name).cpu().numpy(), o_acc[name]):
                    raise AssertionError(f"accumulator plane {name} (and the envelope flag is clear)")
            check_final(fin2, o_fin, "two-step finalize")
            check_final(fin1, o_fin, "fused estimate")
            n["reduce"] += 1
            # ---- round 5: the cohort streamed in chunks through a table taken as streaming state (what the command lines launch); the
            # compact kernel where its shape applies, one chunk now and then through the general kernel; the last chunk finalises, or
            # (a shard) goes slice-major, in which case the buffers must be error_reduce_sliced's over the whole cohort ----
            if S >= 2 and rng.random() < 0.6:
                ncut = int(rng.integers(1, min(S, 4) + 1))
                cuts = [0] + sorted(rng.choice(np.arange(1, S), size=ncut - 1, replace=False).tolist()) + [S] if ncut > 1 else [0, S]
                sliced_last = rng.random() < 0.4
                nsl = int(rng.integers(1, 5))
                tacc = ctx.new_acc(P) if len(cuts) > 2 else None
                ctx.set_tuning(1, groups=1)
                Ls, _, _, _ = slice_geometry(P, nsl)
                ws, wg = (torch.zeros(nsl * 21 * Ls, dtype=torch.float64, device="cuda"), torch.zeros(nsl * 8 * Ls, dtype=torch.float32, device="cuda"))
                gs, gg = torch.zeros_like(ws), torch.zeros_like(wg)
                finc = None
                R = P + E
                view = dev.view(S, R, -1)
                for ci in range(len(cuts) - 1):
                    a, b = cuts[ci], cuts[ci + 1]
                    rec = ctx.records(view[a:b].contiguous(), layout, b - a, E=E, dup_off=d_dup)
                    ctx.set_reduce_compact(bool(rng.random() < 0.8))
                    last = ci == len(cuts) - 2
                    if last and sliced_last:
                        ctx.error_reduce_records_sliced(rec, P, tacc, nsl, gs, gg, C_value, cov, first_sample=a, accumulate=ci > 0)
                    else:
                        finc = ctx.error_reduce_records(rec, P, tacc, C_value, cov, first_sample=a, accumulate=ci > 0, finalize=last, summary=True)
                ctx.set_reduce_compact(True)
                if sliced_last:
                    ctx.set_reduce_compact(False)
                    ctx.error_reduce_sliced(dev, P, nsl, ws, wg, C_value, cov, E=E, dup_off=d_dup)
                    ctx.set_reduce_compact(True)
                    if not (torch.equal(gs.view(torch.int64), ws.view(torch.int64)) and torch.equal(gg.view(torch.int32), wg.view(torch.int32))):
                        raise AssertionError(f"slice-major buffers of a streamed shard (cuts {cuts}, {nsl} slices)")
                else:
                    check_final(finc, o_fin, f"streamed chunks {cuts} through a summary table")
                ctx.set_tuning(0)
                if ctx.flags():
                    raise AssertionError("kernel flags after the streamed chunks")
                n["streamed"] = n.get("streamed", 0) + 1
            # ---- round 5: position ranges inside the library (uint16, listed-once panels split; everything else runs whole behind a join) ----
            if rng.random() < 0.3:
                ctx.set_tuning(1, groups=1)
                ctx.set_ranges(int(rng.integers(2, 5)))
                finr = ctx.error_estimate(dev, P, C_value, cov, E=E, dup_off=d_dup)
                finr = ctx.error_estimate(dev, P, C_value, cov, E=E, dup_off=d_dup, out=finr)
                ctx.set_ranges(1)
                ctx.set_tuning(0)
                check_final(finr, o_fin, "error_estimate over position ranges")
                n["ranges"] = n.get("ranges", 0) + 1
            # ---- sliced merge over a random number of shards ----
            if S >= 2:
                nsh = int(rng.integers(2, min(S, 8) + 1))
                L, _, _, bb = slice_geometry(P, nsh)
                sums, gms = [], []
                for r in range(nsh):
                    a, b = shard_range(S, r, nsh)
                    sm = torch.zeros(nsh * 21 * L, dtype=torch.float64, device="cuda")
                    gm = torch.zeros(nsh * 8 * L, dtype=torch.float32, device="cuda")
                    ctx.error_reduce_sliced(dev[a:b].contiguous(), P, nsh, sm, gm, C_value, cov, E=E, dup_off=d_dup, first_sample=a)
                    sums.append(sm)
                    gms.append(gm)
                total = torch.stack(sums).sum(0).view(nsh, 21 * L)
                blocks = torch.zeros(nsh * bb, dtype=torch.uint8, device="cuda")
                for k in range(nsh):
                    recv = torch.stack([g.view(nsh, 8 * L)[k] for g in gms]).contiguous()
                    ctx.error_finalize_slice(P, nsh, k, total[k].contiguous(), recv, blocks[k * bb:(k + 1) * bb], C_value, cov)
                check_final(ctx.error_table_unslice(P, nsh, blocks), o_fin, f"sliced merge over {nsh} shards")
                n["sliced"] += 1
            # ---- Poisson calling on a tumour cohort of the same panel ----
            T = int(rng.choice([1, 2, 5, 9]))
            trecs = make_records(rng, P, E, T, "edge" if rng.random() < 0.5 else "synth", u16)
            if rng.random() < 0.5:
                trecs[:, :: int(rng.integers(3, 11)), :] = np.array([30, 0, 2, 400, 25, 1, 0, 380], np.int32)
            thr = o_fin["thr"].copy() if rng.random() < 0.5 else rng.choice(
                np.array([0.002, 0.01, 0.0, -1.0, 0.000731, 0.05, 0.3], np.float32), size=(2, 4, P)).astype(np.float32)
            ref_code = rng.integers(0, 4, P).astype(np.uint8)
            ref_code[rng.random(P) < 0.05] = 255
            ext_pos = np.repeat(np.arange(P), mult).astype(np.uint32)
            exp = orc.poisson_call(trecs, P, thr, ref_code, cov, E=E, ext_pos=ext_pos if E else None)
            td, fits = ctx.pack(t(trecs), layout)
            assert fits
            for mode in (POISSON_PREFILTER, POISSON_FULL, POISSON_PREFILTER):
                kw = dict(mode=mode, E=E, ext_pos=t(ext_pos) if E else None, dense_q=(mode == POISSON_FULL),
                          capacity=max(1 << 12, 32 * (P + E) * T * 4))
                ctx.flags()
                ctx.set_ranges(int(rng.integers(2, 5)) if mode == POISSON_PREFILTER and rng.random() < 0.5 else 1)
                res = ctx.poisson_call(td, P, t(thr), t(ref_code), cov, **kw)
                if ctx.flags() & 4:  # more survivors than the default queue holds: size it for the worst case, as the CLI does
                    ctx.set_queue_items((P + E) * T * 3)
                    res = ctx.poisson_call(td, P, t(thr), t(ref_code), cov, **kw)
                    n["queue_regrown"] = n.get("queue_regrown", 0) + 1
                    if ctx.flags() & 4:
                        raise AssertionError("queue overflow persists after growing the queue")
                if not np.array_equal(res["call_mask"].cpu().numpy(), exp["call_mask"]):
                    got = res["call_mask"].cpu().numpy()
                    bad = np.argwhere(got != exp["call_mask"])
                    print("flags", ctx.flags(clear=False), "differing records", len(bad), "first", bad[:5].tolist(),
                          [(int(got[a, b]), int(exp["call_mask"][a, b]), trecs[a, b].tolist(), thr[:, :, b if b < P else ext_pos[b - P]].tolist(),
                            int(ref_code[b if b < P else ext_pos[b - P]])) for a, b in bad[:3]], flush=True)
                    raise AssertionError(f"call mask, mode {mode}")
                if mode == POISSON_FULL:
                    q = res["q"].cpu().numpy()
                    if not (np.array_equal(q == -1, exp["q"] == -1) and np.max(np.abs(q - exp["q"])) <= 1e-5):
                        raise AssertionError("dense Q")
                calls = ctx.read_calls(res)
                ctx.set_ranges(1)
                if len(calls) != sum(bin(int(v)).count("1") for v in exp["call_mask"].ravel()):
                    raise AssertionError(f"call list length, mode {mode}")
            n["poisson"] += 1
        except AssertionError as e:
            print(f"MISMATCH in case seed={seed}: {e}  (P={P} S={S} E={E} layout={layout} kind={kind} C={C_value} cov={cov})", flush=True)
            sys.exit(1)
        if case % 50 == 0:
            print(f"{case} cases ok {n}", flush=True)
    print(f"fuzz parity: {case} cases, all equal to the oracle {n}", flush=True)


if __name__ == "__main__":
    main()
