"""numpy/ctypes binding of oracle/_build/liboracle.so -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this.  The product (amplisolve_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "liboracle.so")
REF_EE_DRIVER = os.path.join(HERE, "_ref", "ee_ref_driver")
REF_VC_SCORER = os.path.join(HERE, "_ref", "libvc_scorer_ref.so")
REF_VC_DRIVER = os.path.join(HERE, "_ref", "vc_ref_driver")
# "callVariants without its Fisher statements" (oracle/Makefile, VC_CALL_DROP): the reference's command line / its phases with a clock
REF_VC_NOFISHER = os.path.join(HERE, "_ref", "AmpliSolveVariantCalling_noFisher")
REF_VC_CALL_DRIVER = os.path.join(HERE, "_ref", "vc_call_ref_driver")

_lib = None


def build():
    subprocess.run(["make", "-s", "-C", HERE], check=True)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.oracle_kf_lgamma.restype = C.c_double
        L.oracle_kf_lgamma.argtypes = [C.c_double]
        L.oracle_kf_gammaq.restype = C.c_double
        L.oracle_kf_gammaq.argtypes = [C.c_double, C.c_double]
        L.oracle_pvalue.restype = C.c_double
        L.oracle_pvalue.argtypes = [C.c_int, C.c_int, C.c_float]
        L.oracle_score.restype = C.c_longdouble
        L.oracle_score.argtypes = [C.c_int, C.c_int, C.c_float]
        L.oracle_text_roundtrip.restype = C.c_float
        L.oracle_text_roundtrip.argtypes = [C.c_float]
        L.oracle_af_gate.restype = C.c_int
        L.oracle_af_gate.argtypes = [C.c_int32, C.c_int32]
        L.oracle_format_thr_cell.restype = C.c_int
        L.oracle_format_thr_cell.argtypes = [C.c_uint8, C.c_float, C.c_float, C.c_int, C.c_char_p]
        L.oracle_format_germ_cell.restype = C.c_int
        L.oracle_format_germ_cell.argtypes = [C.c_uint8, C.c_double, C.c_char_p]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else C.c_void_p(0)


def score_batch(k, rd, err):
    k = np.ascontiguousarray(k, np.int32)
    rd = np.ascontiguousarray(rd, np.int32)
    err = np.ascontiguousarray(err, np.float32)
    q = np.empty(k.size, np.float64)
    p = np.empty(k.size, np.float64)
    lib().oracle_score_batch(_p(k), _p(rd), _p(err), C.c_int64(k.size), _p(q), _p(p))
    return q, p


def error_reduce(recs, P, C_value=0.002, cov=100, E=0, dup_off=None, first_sample=0, rd=None):
    """rd: optional int32 [S][P+E] RD column, INT32_MIN where the line's RD equals A+C+G+T"""
    recs = np.ascontiguousarray(recs, np.int32)
    S = recs.shape[0]
    assert recs.size == S * (P + E) * 8
    if rd is not None:
        rd = np.ascontiguousarray(rd, np.int32)
        assert rd.size == S * (P + E)
    out = dict(snt=np.empty((2, 4, P), np.float64), srd=np.empty((2, 4, P), np.int64), cnt=np.empty((4, P), np.int32),
               nrec=np.empty((P,), np.int32), gm_n=np.empty((4, P), np.int32), gm_first=np.empty((4, P), np.int32),
               gm_first_af=np.empty((4, P), np.float32), gm_rest=np.empty((4, P), np.float32))
    flag = C.c_int32(0)
    if dup_off is not None:
        dup_off = np.ascontiguousarray(dup_off, np.uint32)
    lib().oracle_error_reduce_rd(_p(recs), _p(rd), C.c_int64(P), C.c_int64(E), _p(dup_off), C.c_int32(S), C.c_int32(first_sample),
                              C.c_float(C_value), C.c_int32(cov), _p(out["snt"]), _p(out["srd"]), _p(out["cnt"]),
                              _p(out["nrec"]), _p(out["gm_n"]), _p(out["gm_first"]), _p(out["gm_first_af"]),
                              _p(out["gm_rest"]), C.byref(flag))
    out["order_sensitive"] = int(flag.value)
    return out


def error_sums_inorder(recs, P, C_value=0.002, cov=100, E=0, dup_off=None, rd=None):
    """snt [2, 4, P] in the reference's own order of addition (oracle_error_sums_inorder): what a cohort outside the exactness
    envelope (error_reduce(...)["order_sensitive"] == 1) gets in the reference's table"""
    recs = np.ascontiguousarray(recs, np.int32)
    S = recs.shape[0]
    assert recs.size == S * (P + E) * 8
    if rd is not None:
        rd = np.ascontiguousarray(rd, np.int32)
    if dup_off is not None:
        dup_off = np.ascontiguousarray(dup_off, np.uint32)
    snt = np.empty((2, 4, P), np.float64)
    lib().oracle_error_sums_inorder(_p(recs), _p(rd), C.c_int64(P), C.c_int64(E), _p(dup_off), C.c_int32(S), C.c_float(C_value), C.c_int32(cov), _p(snt))
    return snt


def acc_merge(L, R):
    """L (+) R in place on copies; L covers the earlier samples."""
    L = {k: (v.copy() if isinstance(v, np.ndarray) else v) for k, v in L.items()}
    P = L["nrec"].shape[0]
    lib().oracle_acc_merge(C.c_int64(P), _p(L["snt"]), _p(L["srd"]), _p(L["cnt"]), _p(L["nrec"]), _p(L["gm_n"]),
                           _p(L["gm_first"]), _p(L["gm_first_af"]), _p(L["gm_rest"]), _p(R["snt"]), _p(R["srd"]),
                           _p(R["cnt"]), _p(R["nrec"]), _p(R["gm_n"]), _p(R["gm_first"]), _p(R["gm_first_af"]),
                           _p(R["gm_rest"]))
    return L


def error_finalize(acc):
    P = acc["nrec"].shape[0]
    out = dict(rate=np.empty((2, 4, P), np.float32), code=np.empty((4, P), np.uint8), thr=np.empty((2, 4, P), np.float32),
               germ_val=np.empty((4, P), np.float64), germ_present=np.empty((4, P), np.uint8))
    lib().oracle_error_finalize(C.c_int64(P), _p(acc["snt"]), _p(acc["srd"]), _p(acc["cnt"]), _p(acc["nrec"]),
                                _p(acc["gm_n"]), _p(acc["gm_rest"]), _p(out["rate"]), _p(out["code"]), _p(out["thr"]),
                                _p(out["germ_val"]), _p(out["germ_present"]))
    return out


def poisson_call(trecs, P, thr, ref_code, cov=100, E=0, ext_pos=None, dense=True, rd=None):
    """rd: optional int32 [T][P+E] RD column, INT32_MIN where the line's RD equals A+C+G+T"""
    trecs = np.ascontiguousarray(trecs, np.int32)
    T = trecs.shape[0]
    R = P + E
    assert trecs.size == T * R * 8
    if rd is not None:
        rd = np.ascontiguousarray(rd, np.int32)
        assert rd.size == T * R
    thr = np.ascontiguousarray(thr, np.float32)
    ref_code = np.ascontiguousarray(ref_code, np.uint8)
    mask = np.empty((T, R), np.uint8)
    q = np.empty((T, R, 4, 2), np.float64) if dense else None
    af = np.empty((T, R, 4, 3), np.float32) if dense else None
    if ext_pos is not None:
        ext_pos = np.ascontiguousarray(ext_pos, np.uint32)
    lib().oracle_poisson_call_rd(_p(trecs), _p(rd), C.c_int64(P), C.c_int64(E), _p(ext_pos), C.c_int32(T), _p(thr), _p(ref_code),
                              C.c_int32(cov), _p(mask), _p(q), _p(af))
    return dict(call_mask=mask, q=q, af=af)


def thr_cell(code, r_fw, r_bw, is_ref):
    b = C.create_string_buffer(128)
    lib().oracle_format_thr_cell(int(code), float(r_fw), float(r_bw), int(is_ref), b)
    return b.value.decode()


def germ_cell(present, v):
    b = C.create_string_buffer(128)
    lib().oracle_format_germ_cell(int(present), float(v), b)
    return b.value.decode()


def format_error_table_rows(fin, ref_code, chrom_pos, dup_flags):
    """Rows of positionSpecificNoise_*.txt (EE:2561, EE:2654-2849) for unique positions, in the given order."""
    rows = []
    P = len(chrom_pos)
    for p in range(P):
        chrom, pos, refb = chrom_pos[p]
        cells = [chrom, str(pos), refb, "YES" if dup_flags[p] else "NO"]
        for nt in range(4):
            cells.append(thr_cell(fin["code"][nt, p], fin["rate"][0, nt, p], fin["rate"][1, nt, p], "ACGT"[nt] == refb))
        for nt in range(4):
            cells.append(germ_cell(fin["germ_present"][nt, p], fin["germ_val"][nt, p]))
        rows.append("\t".join(cells))
    return rows
