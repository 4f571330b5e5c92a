"""Thin Python harness over the C ABI of libamplisolve_hip.so.

torch is used only for plumbing: device memory (tensors), the current HIP
stream and torch.distributed.  Every number comes out of the HIP kernels; when
the library or the GPU is missing the calls raise AmpliError.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

from ._lib import AccTable, AmpliError, AmpliNoDevice, Call, Records, hip_lib

NT = "ACGT"
POISSON_FULL = 0
POISSON_PREFILTER = 1
CALL_SHARDS = 32            # AMPLI_CALL_SHARDS
CALL_COUNTER_STRIDE = 16    # AMPLI_CALL_COUNTER_STRIDE
CALL_COUNTER_WORDS = CALL_SHARDS * CALL_COUNTER_STRIDE


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class Acc:
    """Accumulator table (ampli_acc_table) living in one device buffer."""

    def __init__(self, ctx: "Context | None", P: int, buf=None, device=None):
        """ctx may be None for a host-side table (layout functions need no GPU): pass device="cpu"."""
        import torch

        lib = ctx.lib if ctx is not None else hip_lib()
        self.P = int(P)
        nbytes = lib.ampli_acc_bytes(self.P)
        dev = device if device is not None else ctx.device
        if buf is None:
            # over-allocate on the host so the base can be 256-byte aligned like a device allocation
            raw = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
            off = (-raw.data_ptr()) % 256
            buf = raw[off: off + nbytes]
        self.buf = buf
        assert self.buf.numel() >= nbytes and self.buf.data_ptr() % 256 == 0
        self.struct = AccTable()
        rc = lib.ampli_acc_bind(_ptr(self.buf), self.P, C.byref(self.struct))
        if rc != 0:
            raise AmpliError("ampli_acc_bind failed")
        base = self.buf.data_ptr()

        def view(ptr, dtype, shape):
            n = 1
            for d in shape:
                n *= d
            off = ptr - base
            return self.buf[off: off + n * torch.empty(0, dtype=dtype).element_size()].view(dtype).view(*shape)

        s = self.struct
        P_ = self.P
        self.snt = view(s.snt, torch.float64, (2, 4, P_))
        self.srd = view(s.srd, torch.int64, (2, 4, P_))
        self.cnt = view(s.cnt, torch.int32, (4, P_))
        self.nrec = view(s.nrec, torch.int32, (P_,))
        self.gm_n = view(s.gm_n, torch.int32, (4, P_))
        self.gm_first = view(s.gm_first, torch.int32, (4, P_))
        self.gm_first_af = view(s.gm_first_af, torch.float32, (4, P_))
        self.gm_rest = view(s.gm_rest, torch.float32, (4, P_))

    def planes(self):
        return dict(snt=self.snt, srd=self.srd, cnt=self.cnt, nrec=self.nrec, gm_n=self.gm_n,
                    gm_first=self.gm_first, gm_first_af=self.gm_first_af, gm_rest=self.gm_rest)


@dataclass
class ErrorTable:
    rate: "object"          # [2,4,P] float32
    code: "object"          # [4,P] uint8
    thr: "object"           # [2,4,P] float32 (after the text round trip)
    germ_val: "object"      # [4,P] float32
    germ_present: "object"  # [4,P] uint8
    flags: "object"         # [1] int32


class Context:
    def __init__(self, device: int = 0, own_stream: bool = False):
        import torch

        self.lib = hip_lib()
        # two different probes, two different messages (a failure here once could not be told apart afterwards)
        msg = C.create_string_buffer(256)
        n = self.lib.ampli_device_probe(msg, len(msg))
        if n <= 0:
            why = msg.value.decode() if n < 0 else "hipGetDeviceCount succeeded and counted 0 devices"
            raise AmpliNoDevice(f"no MI355X visible to libamplisolve_hip.so ({why}); there is no CPU fallback")
        if not torch.cuda.is_available():
            raise AmpliError(f"libamplisolve_hip.so sees {n} device(s) but torch.cuda.is_available() is False "
                             f"(torch {torch.__version__}, hip {getattr(torch.version, 'hip', None)}): the tensors this API hands out need torch's runtime")
        self.device = torch.device("cuda", device)
        torch.cuda.set_device(self.device)
        stream = C.c_void_p(-1) if own_stream else C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        h = C.c_void_p()
        rc = self.lib.ampli_ctx_create(device, stream, C.byref(h))
        if rc != 0:
            raise AmpliError(f"ampli_ctx_create: {self.lib.ampli_strerror(rc).decode()}")
        self.h = h
        self.own_stream = own_stream

    def close(self):
        if getattr(self, "h", None):
            self.lib.ampli_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc != 0:
            detail = self.lib.ampli_last_error(self.h).decode() if getattr(self, "h", None) else ""
            raise AmpliError(f"{self.lib.ampli_strerror(rc).decode()} ({rc}): {detail}")

    def sync(self):
        self._check(self.lib.ampli_sync(self.h))

    # ---- record layout (include/amplisolve_hip.h: AMPLI_RECORDS_I32 / AMPLI_RECORDS_U16) ----
    LAYOUTS = {"i32": 0, "u16": 1, "u24": 2}

    def _rec_dtype(self):
        import torch

        return {"i32": torch.int32, "u16": torch.int16, "u24": torch.uint8}[getattr(self, "_layout", "i32")]

    def _rec_elems(self) -> int:
        """elements of _rec_dtype per record"""
        return 24 if getattr(self, "_layout", "i32") == "u24" else 8

    def set_record_layout(self, layout):
        """Every record tensor handed to this context from now on is int32 [S, R, 8] ("i32"), int16 [S, R, 8] ("u16")
        or uint8 [S, R, 24] ("u24": 8 little-endian 24-bit fields).  True / False select "u16" / "i32"."""
        if isinstance(layout, bool):
            layout = "u16" if layout else "i32"
        self._check(self.lib.ampli_set_record_layout(self.h, self.LAYOUTS[layout]))
        self._layout = layout

    def pack16(self, recs):
        """int32 [S, R, 8] records -> (int16 [S, R, 8] records, fits): fits is False when a count exceeds 65534."""
        import torch

        assert recs.dtype == torch.int32 and recs.is_cuda and recs.is_contiguous() and recs.shape[-1] == 8
        out = torch.empty(recs.shape, dtype=torch.int16, device=recs.device)
        over = torch.zeros((1,), dtype=torch.int32, device=recs.device)
        self._check(self.lib.ampli_records_pack16(self.h, _ptr(recs), recs.numel() // 8, _ptr(out), _ptr(over)))
        return out, int(over.item()) == 0

    def pack24(self, recs):
        """int32 [S, R, 8] records -> (uint8 [S, R, 24] records, fits): fits is False when a count exceeds 2^24 - 2."""
        import torch

        assert recs.dtype == torch.int32 and recs.is_cuda and recs.is_contiguous() and recs.shape[-1] == 8
        out = torch.empty(tuple(recs.shape[:-1]) + (24,), dtype=torch.uint8, device=recs.device)
        over = torch.zeros((1,), dtype=torch.int32, device=recs.device)
        self._check(self.lib.ampli_records_pack24(self.h, _ptr(recs), recs.numel() // 8, _ptr(out), _ptr(over)))
        return out, int(over.item()) == 0

    def pack(self, recs, layout: str):
        """recs (int32) in `layout`: (tensor, fits)"""
        return (recs, True) if layout == "i32" else (self.pack16(recs) if layout == "u16" else self.pack24(recs))

    def set_tuning(self, reduce_splits: int = 0, general: bool = False, groups: int = 0):
        self._check(self.lib.ampli_set_tuning(self.h, reduce_splits, int(general), groups))

    def last_reduce_kernel(self) -> str:
        """the kernel the latest error_reduce launch of this context was"""
        k = self.lib.ampli_last_reduce_kernel(self.h)
        self._check(min(k, 0))
        return {1: "error_reduce_u16_kernel", 2: "error_reduce_u24_kernel"}.get(k, "error_reduce_kernel")

    # ---- hipGraph capture ---------------------------------------------------------------
    def graph_begin(self):
        self._check(self.lib.ampli_graph_begin(self.h))

    def graph_end(self):
        g = C.c_void_p()
        self._check(self.lib.ampli_graph_end(self.h, C.byref(g)))
        return g

    def graph_launch(self, g):
        self._check(self.lib.ampli_graph_launch(self.h, g))

    def graph_destroy(self, g):
        self.lib.ampli_graph_destroy(g)

    def set_async_drain(self, on: bool):
        """poisson_call's drain kernel on a side stream; results complete after wait_calls() / sync()."""
        self._check(self.lib.ampli_set_async_drain(self.h, int(on)))

    def wait_calls(self):
        self._check(self.lib.ampli_wait_calls(self.h))

    def set_ranges(self, n: int):
        """error_estimate / poisson_call (prefilter) over n tile-aligned position ranges on n streams (include/amplisolve_hip.h,
        ampli_set_ranges); 1 = off.  The section closes at the next other call on this context."""
        self._check(self.lib.ampli_set_ranges(self.h, n))

    def ranges_concurrent(self) -> bool:
        """did ampli_set_ranges see every pair of the ranges' streams overlap (different hardware queues)?"""
        return self.lib.ampli_ranges_concurrent(self.h) == 1

    def ranges_join(self):
        self._check(self.lib.ampli_ranges_join(self.h))

    def range_record(self, rng: int, ev):
        """record `ev` on range `rng`'s stream without closing the section (ampli_range_event_record)"""
        self._check(self.lib.ampli_range_event_record(self.h, rng, ev))

    def set_reduce_compact(self, on):
        """True / False; 2 = the compact-state kernel for uint16 records only (A/B runs of its 24-bit form)"""
        self._check(self.lib.ampli_set_reduce_compact(self.h, 2 if on == 2 else int(bool(on))))

    def set_slice_format(self, slim: bool):
        """sums of the sliced exchange as 14 packed planes (slim) or 21 plain ones (ampli_set_slice_format)"""
        self._check(self.lib.ampli_set_slice_format(self.h, 1 if slim else 0))

    def set_poisson_tuning(self, rows_per_wave: int = 0, drain_blocks: int = 0):
        """poisson_call launch shape (ampli_set_poisson_tuning); 0 = default.  Results do not depend on it."""
        self._check(self.lib.ampli_set_poisson_tuning(self.h, rows_per_wave, drain_blocks))

    def set_queue_items(self, items: int):
        self._check(self.lib.ampli_set_queue_items(self.h, items))

    def flags(self, clear: bool = True) -> int:
        """AMPLI_FLAG_* raised by kernels since the last clear (synchronises)."""
        out = C.c_int32()
        self._check(self.lib.ampli_ctx_flags(self.h, C.byref(out), int(clear)))
        return int(out.value)

    # ---- events on the context's stream -------------------------------------------------
    def event(self):
        ev = C.c_void_p()
        self._check(self.lib.ampli_event_create(C.byref(ev)))
        return ev

    def record(self, ev):
        self._check(self.lib.ampli_event_record(self.h, ev))

    def elapsed_ms(self, a, b) -> float:
        ms = C.c_float()
        self._check(self.lib.ampli_event_elapsed_ms(a, b, C.byref(ms)))
        return float(ms.value)

    # ---- kernels ---------------------------------------------------------------------
    def new_acc(self, P: int) -> Acc:
        return Acc(self, P)

    def error_reduce(self, recs, P: int, C_value: float = 0.002, cov: int = 100, E: int = 0, dup_off=None,
                     first_sample: int = 0, acc: Acc | None = None) -> Acc:
        import torch

        assert recs.dtype == self._rec_dtype() and recs.is_cuda and recs.is_contiguous()
        S = recs.shape[0]
        assert recs.numel() == S * (P + E) * self._rec_elems()
        if acc is None:
            acc = self.new_acc(P)
        self._check(self.lib.ampli_error_reduce(self.h, _ptr(recs), P, E, _ptr(dup_off), S, first_sample,
                                                C_value, cov, C.byref(acc.struct)))
        return acc

    def records(self, recs, layout: str, n_samples: int, E: int = 0, row_stride: int = 0, ext=None, ext_stride: int = 0,
                dup_off=None, ext_pos=None, rd=None, rd_ext=None) -> Records:
        """ampli_records over device tensors (the caller keeps them alive).  rd / rd_ext: int32 [n][P] / [n][E] RD column of
        lines whose RD is not A+C+G+T (INT32_MIN elsewhere)."""
        r = Records(recs.data_ptr(), row_stride, ext.data_ptr() if ext is not None else None, ext_stride, E,
                    dup_off.data_ptr() if dup_off is not None else None, ext_pos.data_ptr() if ext_pos is not None else None,
                    self.LAYOUTS[layout], n_samples, rd.data_ptr() if rd is not None else None,
                    rd_ext.data_ptr() if rd_ext is not None else None)
        r._keep = (recs, ext, dup_off, ext_pos, rd, rd_ext)
        return r

    def error_reduce_records(self, rec: Records, P: int, acc: Acc | None, C_value: float = 0.002, cov: int = 100, first_sample: int = 0,
                             accumulate: bool = False, out: ErrorTable | None = None, finalize: bool = False, summary: bool = False):
        """One chunk of a cohort into `acc` (accumulate: acc (+) chunk); finalize: also the error table of the merged state.
        summary: `acc` is streaming state only (AMPLI_REDUCE_SUMMARY: gm_n may saturate at two, gm_first may read -1), which lets
        the compact-state kernel carry it."""
        if finalize and out is None:
            out = self._new_error_table(P)
        o = out if finalize else None
        self._check(self.lib.ampli_error_reduce_records(self.h, C.byref(rec), P, first_sample, C_value, cov,
                                                        C.byref(acc.struct) if acc is not None else None, int(bool(accumulate)) | (2 if summary else 0),
                                                        _ptr(o.rate) if o else None, _ptr(o.code) if o else None, _ptr(o.thr) if o else None,
                                                        _ptr(o.germ_val) if o else None, _ptr(o.germ_present) if o else None,
                                                        _ptr(o.flags) if o else None))
        return out

    def error_sums_inorder(self, rec: Records, P: int, acc: Acc, C_value: float = 0.002, cov: int = 100, accumulate: bool = False):
        """acc.snt in the reference's own order of addition (ampli_error_sums_inorder): one chunk, last sample first; a cohort in several
        chunks from its LAST chunk (accumulate=False) to its first (True).  The other planes of acc are left alone."""
        self._check(self.lib.ampli_error_sums_inorder(self.h, C.byref(rec), P, C_value, cov, C.byref(acc.struct), int(bool(accumulate))))

    def error_reduce_records_sliced(self, rec: Records, P: int, acc: Acc | None, n_slices: int, sums, gm, C_value: float = 0.002, cov: int = 100,
                                    first_sample: int = 0, accumulate: bool = False, summary: bool = True):
        """The last chunk of a shard's streamed cohort: acc (+) chunk straight into the slice-major exchange buffers."""
        self._check(self.lib.ampli_error_reduce_records_sliced(self.h, C.byref(rec), P, first_sample, C_value, cov,
                                                               C.byref(acc.struct) if acc is not None else None,
                                                               int(bool(accumulate)) | (2 if summary else 0), n_slices, _ptr(sums), _ptr(gm)))

    def poisson_call_records(self, rec: Records, P: int, thr, ref_code, cov: int = 100, mode: int = POISSON_PREFILTER,
                             call_mask=None, capacity: int = 0, calls_buf=None, n_calls=None):
        import torch

        T, R = rec.n_samples, P + rec.E
        d = self.device
        if call_mask is None:
            call_mask = torch.empty(((T * R + 3) // 4 * 4,), dtype=torch.uint8, device=d)[: T * R].view(T, R)
        if capacity > 0 and calls_buf is None:
            calls_buf = torch.empty((capacity * C.sizeof(Call),), dtype=torch.uint8, device=d)
        if capacity > 0:
            capacity -= capacity % CALL_SHARDS
        if (capacity > 0 or n_calls is not None) and n_calls is None:
            n_calls = torch.zeros((CALL_COUNTER_WORDS,), dtype=torch.int64, device=d)
        self._check(self.lib.ampli_poisson_call_records(self.h, C.byref(rec), P, _ptr(thr), _ptr(ref_code), cov, mode, _ptr(call_mask),
                                                        _ptr(calls_buf), capacity, _ptr(n_calls), None, None))
        return dict(call_mask=call_mask, q=None, af=None, calls_buf=calls_buf, n_calls=n_calls, capacity=capacity)

    def _new_error_table(self, P: int) -> ErrorTable:
        import torch

        d = self.device
        return ErrorTable(rate=torch.empty((2, 4, P), dtype=torch.float32, device=d),
                          code=torch.empty((4, P), dtype=torch.uint8, device=d),
                          thr=torch.empty((2, 4, P), dtype=torch.float32, device=d),
                          germ_val=torch.empty((4, P), dtype=torch.float32, device=d),
                          germ_present=torch.empty((4, P), dtype=torch.uint8, device=d),
                          flags=torch.zeros((1,), dtype=torch.int32, device=d))

    def error_estimate(self, recs, P: int, C_value: float = 0.002, cov: int = 100, E: int = 0, dup_off=None,
                       acc: Acc | None = None, out: ErrorTable | None = None) -> ErrorTable:
        """Fused error_reduce + error_finalize (ampli_error_estimate); acc optional."""
        import torch

        assert recs.dtype == self._rec_dtype() and recs.is_cuda and recs.is_contiguous()
        S = recs.shape[0]
        assert recs.numel() == S * (P + E) * self._rec_elems()
        if out is None:
            out = self._new_error_table(P)
        self._check(self.lib.ampli_error_estimate(self.h, _ptr(recs), P, E, _ptr(dup_off), S, C_value, cov,
                                                  C.byref(acc.struct) if acc is not None else None, _ptr(out.rate), _ptr(out.code),
                                                  _ptr(out.thr), _ptr(out.germ_val), _ptr(out.germ_present), _ptr(out.flags)))
        return out

    def error_reduce_packed(self, recs, P: int, acc: Acc, packed, C_value: float = 0.002, cov: int = 100, E: int = 0,
                            dup_off=None, first_sample: int = 0):
        """Shard reduction for the multi-GPU merge: additive planes -> packed (float64 [21*P]), gm planes -> acc."""
        S = recs.shape[0]
        self._check(self.lib.ampli_error_reduce_packed(self.h, _ptr(recs), P, E, _ptr(dup_off), S, first_sample, C_value, cov,
                                                       C.byref(acc.struct), _ptr(packed)))

    def error_finalize_merged(self, P: int, packed, gm_regions, nparts: int, C_value: float = 0.002, cov: int = 100,
                              out: ErrorTable | None = None) -> ErrorTable:
        if out is None:
            out = self._new_error_table(P)
        self._check(self.lib.ampli_error_finalize_merged(self.h, P, _ptr(packed), _ptr(gm_regions), nparts, C_value, cov,
                                                         _ptr(out.rate), _ptr(out.code), _ptr(out.thr), _ptr(out.germ_val),
                                                         _ptr(out.germ_present), _ptr(out.flags)))
        return out

    # ---- position-sliced merge (reduce-scatter / all-to-all / all-gather; include/amplisolve_hip.h) ----
    def set_slice_group(self, group_size: int = 1, group_index: int = 0):
        """The sliced exchange buffers hold group_size batches per slice chunk; the following sliced calls address batch
        group_index (include/amplisolve_hip.h, ampli_set_slice_group)."""
        self._check(self.lib.ampli_set_slice_group(self.h, group_size, group_index))

    def slice_len(self, P: int, n_slices: int) -> int:
        return int(self.lib.ampli_slice_len(P, n_slices))

    def error_reduce_sliced(self, recs, P: int, n_slices: int, sums, gm, C_value: float = 0.002, cov: int = 100, E: int = 0,
                            dup_off=None, first_sample: int = 0):
        """Shard reduction straight into the exchange buffers: sums f64 [n][21][L], gm f32 [n][8][L]."""
        S = recs.shape[0]
        self._check(self.lib.ampli_error_reduce_sliced(self.h, _ptr(recs), P, E, _ptr(dup_off), S, first_sample, C_value, cov,
                                                       n_slices, _ptr(sums), _ptr(gm)))

    def error_finalize_slice(self, P: int, n_slices: int, slice_index: int, sum_slice, gm_recv, block, C_value: float = 0.002,
                             cov: int = 100):
        self._check(self.lib.ampli_error_finalize_slice(self.h, P, n_slices, slice_index, _ptr(sum_slice), _ptr(gm_recv),
                                                        C_value, cov, _ptr(block)))

    def error_table_unslice(self, P: int, n_slices: int, blocks, out: ErrorTable | None = None) -> ErrorTable:
        if out is None:
            out = self._new_error_table(P)
        self._check(self.lib.ampli_error_table_unslice(self.h, P, n_slices, _ptr(blocks), _ptr(out.rate), _ptr(out.code),
                                                       _ptr(out.thr), _ptr(out.germ_val), _ptr(out.germ_present), _ptr(out.flags)))
        return out

    def acc_merge(self, parts: list[Acc], dst: Acc | None = None) -> Acc:
        if dst is None:
            dst = self.new_acc(parts[0].P)
        arr = (AccTable * len(parts))(*[p.struct for p in parts])
        self._check(self.lib.ampli_acc_merge(self.h, C.byref(dst.struct), arr, len(parts)))
        return dst

    def regions(self, P: int):
        """(sum_bytes, gm_offset, gm_bytes) of a table buffer (ampli_acc_regions)."""
        a, b, c = C.c_size_t(), C.c_size_t(), C.c_size_t()
        self._check(self.lib.ampli_acc_regions(P, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def acc_pack(self, acc: Acc, packed):
        self._check(self.lib.ampli_acc_pack(self.h, C.byref(acc.struct), _ptr(packed)))

    def acc_unpack(self, packed, acc: Acc):
        self._check(self.lib.ampli_acc_unpack(self.h, _ptr(packed), C.byref(acc.struct)))

    def gm_merge(self, dst: Acc, regions, nparts: int):
        """regions: uint8 tensor holding nparts gathered gm regions back to back."""
        self._check(self.lib.ampli_gm_merge(self.h, C.byref(dst.struct), _ptr(regions), nparts))

    def error_finalize(self, acc: Acc, C_value: float = 0.002, cov: int = 100, out: ErrorTable | None = None) -> ErrorTable:
        import torch

        P = acc.P
        if out is None:
            out = self._new_error_table(P)
        self._check(self.lib.ampli_error_finalize(self.h, C.byref(acc.struct), C_value, cov, _ptr(out.rate), _ptr(out.code),
                                                  _ptr(out.thr), _ptr(out.germ_val), _ptr(out.germ_present), _ptr(out.flags)))
        return out

    def poisson_call(self, trecs, P: int, thr, ref_code, cov: int = 100, mode: int = POISSON_PREFILTER, E: int = 0,
                     ext_pos=None, call_mask=None, capacity: int = 0, dense_q: bool = False, dense_af: bool = False,
                     calls_buf=None, n_calls=None, blocks_of: int = 0):
        """thr: float32 [2, 4, P] thresholds -- or, with blocks_of = n > 0, the all-gathered blocks of an n-slice
        position-sliced merge (ampli_poisson_call_blocks: thresholds are read straight from the blocks)."""
        import torch

        assert trecs.dtype == self._rec_dtype() and trecs.is_cuda and trecs.is_contiguous()
        T = trecs.shape[0]
        R = P + E
        assert trecs.numel() == T * R * self._rec_elems()
        d = self.device
        if call_mask is None:
            call_mask = torch.empty(((T * R + 3) // 4 * 4,), dtype=torch.uint8, device=d)[: T * R].view(T, R)
        q = torch.empty((T, R, 4, 2), dtype=torch.float64, device=d) if dense_q else None
        af = torch.empty((T, R, 4, 3), dtype=torch.float32, device=d) if dense_af else None
        if capacity > 0 and calls_buf is None:
            calls_buf = torch.empty((capacity * C.sizeof(Call),), dtype=torch.uint8, device=d)
        if capacity > 0:
            capacity -= capacity % CALL_SHARDS
        if (capacity > 0 or n_calls is not None) and n_calls is None:
            n_calls = torch.zeros((CALL_COUNTER_WORDS,), dtype=torch.int64, device=d)
        if blocks_of > 0:
            self._check(self.lib.ampli_poisson_call_blocks(self.h, _ptr(trecs), P, E, _ptr(ext_pos), T, _ptr(thr), blocks_of,
                                                           _ptr(ref_code), cov, mode, _ptr(call_mask), _ptr(calls_buf), capacity,
                                                           _ptr(n_calls), _ptr(q), _ptr(af)))
        else:
            self._check(self.lib.ampli_poisson_call(self.h, _ptr(trecs), P, E, _ptr(ext_pos), T, _ptr(thr), _ptr(ref_code), cov,
                                                    mode, _ptr(call_mask), _ptr(calls_buf), capacity, _ptr(n_calls), _ptr(q), _ptr(af)))
        return dict(call_mask=call_mask, q=q, af=af, calls_buf=calls_buf, n_calls=n_calls, capacity=capacity)

    @staticmethod
    def n_calls_total(res) -> int:
        return int(res["n_calls"][::CALL_COUNTER_STRIDE].sum().item())

    def read_calls(self, res) -> list[dict]:
        """Copy the compact call list to the host, sorted into the reference's emission order."""
        import numpy as np

        counts = res["n_calls"][::CALL_COUNTER_STRIDE].cpu().numpy()
        per = res["capacity"] // CALL_SHARDS
        if (counts > per).any():
            raise AmpliError(f"call list segment overflowed: {int(counts.max())} > {per}; rerun with a larger capacity")
        sz = C.sizeof(Call)
        raw = b"".join(res["calls_buf"][k * per * sz: (k * per + int(counts[k])) * sz].cpu().numpy().tobytes() for k in range(CALL_SHARDS))
        dt = np.dtype([("sample", "<i4"), ("record", "<i4"), ("alt", "<i4"), ("rd", "<i4"), ("q_fw", "<f8"),
                       ("q_bw", "<f8"), ("af", "<f4"), ("af_fw", "<f4"), ("af_bw", "<f4"), ("k_fw", "<i4"), ("k_bw", "<i4"),
                       ("fw", "<i4"), ("bw", "<i4"), ("flags", "<i4")])
        assert dt.itemsize == sz
        a = np.frombuffer(raw, dtype=dt)
        a = a[np.lexsort((a["alt"], a["record"], a["sample"]))]
        return a

    def score_batch(self, k, rd, err):
        import torch

        n = k.numel()
        q = torch.empty(n, dtype=torch.float64, device=self.device)
        p = torch.empty(n, dtype=torch.float64, device=self.device)
        self._check(self.lib.ampli_score_batch(self.h, _ptr(k), _ptr(rd), _ptr(err), n, _ptr(q), _ptr(p)))
        return q, p

    def score_dense_batch(self, k, rd, err):
        """Q by the all-scores mode's scorer (ampli_poisson_score_dense)"""
        import torch

        q = torch.empty(k.numel(), dtype=torch.float64, device=self.device)
        self._check(self.lib.ampli_score_dense_batch(self.h, _ptr(k), _ptr(rd), _ptr(err), k.numel(), _ptr(q)))
        return q

    def roundtrip_batch(self, x):
        import torch

        out = torch.empty_like(x)
        self._check(self.lib.ampli_roundtrip_batch(self.h, _ptr(x), x.numel(), _ptr(out)))
        return out

    def synth_fill(self, P: int, n_samples: int, first_sample: int = 0, seed: int = 0xA3F15017, depth: int = 2000,
                   tumour: bool = False, out=None):
        import torch

        if out is None:
            out = torch.empty((n_samples, P, 8), dtype=torch.int32, device=self.device)
        self._check(self.lib.ampli_synth_fill(self.h, _ptr(out), P, n_samples, first_sample, seed, depth, int(tumour)))
        return out

    def synth_ref(self, P: int, seed: int = 0xA3F15017):
        import torch

        out = torch.empty((P,), dtype=torch.uint8, device=self.device)
        self._check(self.lib.ampli_synth_ref(self.h, _ptr(out), P, seed))
        return out
