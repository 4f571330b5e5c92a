"""numpy views over the C++ host library's panel / cohort objects (include/amplisolve_host.h)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import CHUNK_FN, AmpliError, host_lib


def _b(s):
    return s.encode() if isinstance(s, str) else s


class HostCohort:
    """BED (or error table) + a directory of .PILEUP.ASEQ files, packed by the C++ host."""

    def __init__(self, bed_or_table, aseq_dir=None, refbases_file=None, fasta=None, is_error_table=False, threads=0,
                 keep_line_no=False, shard=None):
        """shard = (index, count): parse only that contiguous range of the directory's visit order (dist.shard_range);
        first_sample / total_samples then place it in the whole cohort."""
        lib = host_lib()
        h = C.c_void_p()
        k, n = shard if shard is not None else (0, 1)
        rc = lib.ampli_host_cohort_load_shard(_b(bed_or_table), int(is_error_table), _b(refbases_file) if refbases_file else None,
                                              _b(fasta) if fasta else None, _b(aseq_dir) if aseq_dir else None, threads,
                                              int(keep_line_no), k, n, C.byref(h))
        if rc != 0:
            raise AmpliError(f"ampli_host_cohort_load: {lib.ampli_host_last_error().decode()}")
        self._lib, self.h = lib, h
        self.P = lib.ampli_host_cohort_P(h)
        self.E = lib.ampli_host_cohort_E(h)
        self.S = lib.ampli_host_cohort_S(h)
        self.walk_len = lib.ampli_host_cohort_walk_len(h)
        self.first_sample = lib.ampli_host_cohort_first_sample(h)
        self.total_samples = lib.ampli_host_cohort_total_samples(h)
        R = self.P + self.E

        def arr(ptr, shape, dtype):
            if not ptr or 0 in shape:
                return np.zeros(shape, dtype)
            n = int(np.prod(shape))
            buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr)
            return np.frombuffer(buf, dtype=dtype).reshape(shape)

        self.recs = arr(lib.ampli_host_cohort_recs(h), (self.S, R, 8), np.int32) if self.S else np.zeros((0, R, 8), np.int32)
        self.dup_off = arr(lib.ampli_host_cohort_dup_off(h), (self.P + 1,), np.uint32) if self.S else np.zeros(self.P + 1, np.uint32)
        self.ext_pos = arr(lib.ampli_host_cohort_ext_pos(h), (self.E,), np.uint32)
        ln = lib.ampli_host_cohort_line_no(h)
        self.line_no = arr(ln, (self.S, R), np.int32) if ln else None
        n_irr = C.c_int64()
        ip = lib.ampli_host_cohort_irregular(h, C.byref(n_irr))
        self.irregular = arr(ip, (n_irr.value, 4), np.uint32).astype(np.int64) if n_irr.value else np.zeros((0, 4), np.int64)
        self.ref_code = arr(lib.ampli_host_cohort_ref_code(h), (self.P,), np.uint8)
        self.dup_flag = arr(lib.ampli_host_cohort_dup_flag(h), (self.P,), np.uint8)
        self.names = [lib.ampli_host_cohort_sample_name(h, s).decode() for s in range(self.S)]

    def rd_plane(self):
        """int32 [S][P+E]: the RD column of the lines where it is not A+C+G+T, INT32_MIN elsewhere (None: no such line)"""
        if not len(self.irregular):
            return None
        rd = np.full((self.S, self.P + self.E), np.iinfo(np.int32).min, np.int32)
        for s, r, _, v in self.irregular:
            rd[s, r] = np.int32(np.uint32(v).astype(np.int32)) if v > 0x7FFFFFFF else v
        return rd

    def stats(self):
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        self._lib.ampli_host_cohort_stats(self.h, C.byref(a), C.byref(b), C.byref(c), C.byref(d))
        return dict(lines=a.value, offpanel=b.value, irregular=c.value, malformed=d.value)

    def position(self, p):
        buf = C.create_string_buffer(256)
        coord = C.c_int32()
        self._lib.ampli_host_position(self.h, p, buf, 256, C.byref(coord))
        return buf.value.decode(), coord.value

    def write_error_table(self, rate, code, germ_val, germ_present, path):
        rate = np.ascontiguousarray(rate, np.float32)
        code = np.ascontiguousarray(code, np.uint8)
        germ_val = np.ascontiguousarray(germ_val, np.float32)
        germ_present = np.ascontiguousarray(germ_present, np.uint8)
        rc = self._lib.ampli_host_write_error_table(self.h, rate.ctypes.data_as(C.c_void_p), code.ctypes.data_as(C.c_void_p),
                                                    germ_val.ctypes.data_as(C.c_void_p), germ_present.ctypes.data_as(C.c_void_p), _b(path))
        if rc != 0:
            raise AmpliError(self._lib.ampli_host_last_error().decode())

    def stream_chunks(self, aseq_dir, chunk_bytes=128 << 20, threads=0, keep_line_no=False):
        """The directory as the command lines ingest it (ampli_host_stream_chunks): a list of chunks, each a dict with
        first, n, layout (0 = int32, 2 = 24-bit), E, prim / ext (raw uint8 [n][P or E][record bytes]), dup_off, ext_pos,
        line_prim / line_ext, irregular (int64 [k][4]: sample in chunk, record slot, occurrence, RD)."""
        P = self.P
        out = []

        def cb(user, first, n, layout, P_, E, prim, ext, dup_off, ext_pos, line_prim, line_ext, irr, n_irr):
            rb = {0: 32, 1: 16, 2: 24}[layout]

            def arr(ptr, count, dtype):
                if not ptr or count == 0:
                    return np.zeros(count, dtype)
                return np.frombuffer((C.c_char * (count * np.dtype(dtype).itemsize)).from_address(ptr), dtype=dtype).copy()

            out.append(dict(first=first, n=n, layout=layout, E=E, prim=arr(prim, n * P_ * rb, np.uint8).reshape(n, P_, rb),
                            ext=arr(ext, n * E * rb, np.uint8).reshape(n, E, rb), dup_off=arr(dup_off, P_ + 1, np.uint32),
                            ext_pos=arr(ext_pos, E, np.uint32), line_prim=arr(line_prim, n * P_, np.int32).reshape(n, P_) if line_prim else None,
                            line_ext=arr(line_ext, n * E, np.int32).reshape(n, E) if line_ext else None,
                            irregular=arr(irr, n_irr * 4, np.uint32).reshape(n_irr, 4).astype(np.int64)))
            return 0

        rc = self._lib.ampli_host_stream_chunks(self.h, _b(aseq_dir), threads, int(keep_line_no), chunk_bytes, CHUNK_FN(cb), None)
        if rc != 0:
            raise AmpliError(f"ampli_host_stream_chunks ({rc}): {self._lib.ampli_host_last_error().decode()}")
        return out

    def close(self):
        if self.h:
            self._lib.ampli_host_cohort_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def read_error_table(path):
    """-> (HostCohort-like panel object, thr [2,4,P] float32) as AmpliSolveVariantCalling would read it."""
    lib = host_lib()
    # first pass to learn P
    h = C.c_void_p()
    rc = lib.ampli_host_read_error_table(_b(path), C.byref(h), None, 0)
    if rc != 0:
        raise AmpliError(lib.ampli_host_last_error().decode())
    P = lib.ampli_host_cohort_P(h)
    lib.ampli_host_cohort_free(h)
    thr = np.empty((2, 4, P), np.float32)
    h = C.c_void_p()
    rc = lib.ampli_host_read_error_table(_b(path), C.byref(h), thr.ctypes.data_as(C.c_void_p), thr.size)
    if rc != 0:
        raise AmpliError(lib.ampli_host_last_error().decode())
    ref = np.frombuffer((C.c_char * P).from_address(lib.ampli_host_cohort_ref_code(h)), dtype=np.uint8).copy()
    lib.ampli_host_cohort_free(h)
    return ref, thr


class ErrorTable:
    """A positionSpecificNoise table as AmpliSolveVariantCalling keeps it after storeInputFile (VC:430-576): per unique
    position (first row wins) the reference / duplicate / threshold / germ-max cells, and the sequence context of a call."""

    def __init__(self, path, dummy_vcf=None):
        lib = host_lib()
        h = C.c_void_p()
        rc = lib.ampli_host_read_error_table_vcf(_b(path), _b(dummy_vcf) if dummy_vcf else None, C.byref(h), None, 0)
        if rc != 0:
            raise AmpliError(lib.ampli_host_last_error().decode())
        self._lib, self.h = lib, h
        self.P = lib.ampli_host_cohort_P(h)
        self.thr = np.empty((2, 4, self.P), np.float32)
        h2 = C.c_void_p()
        rc = lib.ampli_host_read_error_table(_b(path), C.byref(h2), self.thr.ctypes.data_as(C.c_void_p), self.thr.size)
        if rc != 0:
            raise AmpliError(lib.ampli_host_last_error().decode())
        lib.ampli_host_cohort_free(h2)
        self.dup = np.frombuffer((C.c_char * max(self.P, 1)).from_address(lib.ampli_host_cohort_dup_flag(h)), dtype=np.uint8)[:self.P].copy()

    def key(self, p):
        buf = C.create_string_buffer(256)
        coord = C.c_int32()
        self._lib.ampli_host_position(self.h, p, buf, 256, C.byref(coord))
        return buf.value.decode(), coord.value

    def cell(self, p, which):
        """which: 0 reference, 1..4 threshold A/C/G/T, 5..8 germ-max A/C/G/T"""
        v = self._lib.ampli_host_table_cell(self.h, p, which)
        return None if v is None else v.decode()

    def context(self, p, sub):
        d, u = C.create_string_buffer(1024), C.create_string_buffer(1024)
        flag = self._lib.ampli_host_context(self.h, p, sub.encode(), d, u, 1024)
        if flag < 0:
            raise AmpliError("ampli_host_context")
        return d.value.decode(), u.value.decode(), flag

    def close(self):
        if self.h:
            self._lib.ampli_host_cohort_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sample_order(aseq_dir):
    lib = host_lib()
    buf = C.create_string_buffer(1 << 20)
    n = lib.ampli_host_sample_order(_b(aseq_dir), buf, 1 << 20)
    if n < 0:
        raise AmpliError(lib.ampli_host_last_error().decode())
    return buf.value.decode().split("\n")[:n]


def unpack_records(raw, layout):
    """raw uint8 [..., record bytes] in a device layout -> int32 [..., 8] of the interchange layout (absent: INT32_MIN in field 0)."""
    raw = np.ascontiguousarray(raw)
    lead = raw.shape[:-1]
    if layout == 0:
        return raw.view(np.int32).reshape(lead + (8,)).copy()
    if layout == 1:
        v = raw.view(np.uint16).reshape(lead + (8,)).astype(np.int32)
        v[..., 0] = np.where(v[..., 0] == 0xFFFF, np.iinfo(np.int32).min, v[..., 0])
        return v
    b = raw.reshape(lead + (8, 3)).astype(np.int32)
    v = b[..., 0] | (b[..., 1] << 8) | (b[..., 2] << 16)
    v[..., 0] = np.where(v[..., 0] == 0xFFFFFF, np.iinfo(np.int32).min, v[..., 0])
    return v
