"""ctypes bindings of the two native libraries.  No arithmetic lives here.

libamplisolve_hip.so is the product's compute path; if it is missing or no
MI355X is visible, every compute call raises -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
HIP_LIB_PATH = os.environ.get("AMPLISOLVE_HIP_LIB") or os.path.join(PKG, "lib", "libamplisolve_hip.so")
HOST_LIB_PATH = os.environ.get("AMPLISOLVE_HOST_LIB") or os.path.join(PKG, "lib", "libamplisolve_host.so")

vp, i32, i64, u64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint64, C.c_float, C.c_size_t


class AmpliError(RuntimeError):
    pass


class AmpliNoDevice(AmpliError):
    """hipGetDeviceCount found nothing (or failed): the message carries HIP's own error name and text"""


class AccTable(C.Structure):
    """mirror of ampli_acc_table (include/amplisolve_hip.h)"""
    _fields_ = [("P", i64), ("snt", vp), ("srd", vp), ("cnt", vp), ("nrec", vp), ("gm_n", vp),
                ("gm_first", vp), ("gm_first_af", vp), ("gm_rest", vp)]


class Call(C.Structure):
    """mirror of ampli_call"""
    _fields_ = [("sample", i32), ("record", i32), ("alt", i32), ("rd", i32), ("q_fw", C.c_double),
                ("q_bw", C.c_double), ("af", f32), ("af_fw", f32), ("af_bw", f32), ("k_fw", i32), ("k_bw", i32),
                ("fw", i32), ("bw", i32), ("flags", i32)]


class Records(C.Structure):
    """mirror of ampli_records: a cohort (or a chunk of a streamed one) on the device, described explicitly"""
    _fields_ = [("recs", vp), ("row_stride", i64), ("ext", vp), ("ext_stride", i64), ("E", i64), ("dup_off", vp),
                ("ext_pos", vp), ("layout", i32), ("n_samples", i32), ("rd", vp), ("rd_ext", vp)]


# every symbol include/amplisolve_hip.h declares: (restype, argtypes)
HIP_SYMBOLS = {
    "ampli_abi_version": (C.c_int, []),
    "ampli_strerror": (C.c_char_p, [C.c_int]),
    "ampli_device_count": (C.c_int, []),
    "ampli_device_probe": (C.c_int, [C.c_char_p, C.c_size_t]),
    "ampli_ctx_create": (C.c_int, [C.c_int, vp, C.POINTER(vp)]),
    "ampli_ctx_destroy": (None, [vp]),
    "ampli_last_error": (C.c_char_p, [vp]),
    "ampli_sync": (C.c_int, [vp]),
    "ampli_stream": (vp, [vp]),
    "ampli_pinned_alloc": (C.c_int, [sz, C.POINTER(vp)]),
    "ampli_pinned_free": (C.c_int, [vp]),
    "ampli_host_register": (C.c_int, [vp, vp, C.c_size_t]),
    "ampli_host_unregister": (C.c_int, [vp]),
    "ampli_dev_alloc": (C.c_int, [vp, sz, C.POINTER(vp)]),
    "ampli_dev_free": (C.c_int, [vp, vp]),
    "ampli_copy_h2d": (C.c_int, [vp, vp, vp, sz]),
    "ampli_copy_d2h": (C.c_int, [vp, vp, vp, sz]),
    "ampli_memset_d": (C.c_int, [vp, vp, C.c_int, sz]),
    "ampli_event_create": (C.c_int, [C.POINTER(vp)]),
    "ampli_event_destroy": (C.c_int, [vp]),
    "ampli_event_record": (C.c_int, [vp, vp]),
    "ampli_event_sync": (C.c_int, [vp]),
    "ampli_event_elapsed_ms": (C.c_int, [vp, vp, C.POINTER(f32)]),
    "ampli_acc_bytes": (sz, [i64]),
    "ampli_acc_bind": (C.c_int, [vp, i64, C.POINTER(AccTable)]),
    "ampli_error_reduce": (C.c_int, [vp, vp, i64, i64, vp, i32, i32, f32, i32, C.POINTER(AccTable)]),
    "ampli_error_estimate": (C.c_int, [vp, vp, i64, i64, vp, i32, f32, i32, vp, vp, vp, vp, vp, vp, vp]),
    "ampli_error_reduce_packed": (C.c_int, [vp, vp, i64, i64, vp, i32, i32, f32, i32, C.POINTER(AccTable), vp]),
    "ampli_error_finalize_merged": (C.c_int, [vp, i64, vp, vp, i32, f32, i32, vp, vp, vp, vp, vp, vp]),
    "ampli_set_record_layout": (C.c_int, [vp, i32]),
    "ampli_records_pack16": (C.c_int, [vp, vp, i64, vp, vp]),
    "ampli_records_pack24": (C.c_int, [vp, vp, i64, vp, vp]),
    "ampli_set_slice_group": (C.c_int, [vp, i32, i32]),
    "ampli_slice_len": (i64, [i64, i32]),
    "ampli_slice_bytes": (C.c_int, [i64, i32, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
    "ampli_slice_bytes_fmt": (C.c_int, [i64, i32, i32, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "ampli_slice_planes": (i32, [i32]),
    "ampli_set_slice_format": (C.c_int, [vp, i32]),
    "ampli_error_reduce_sliced": (C.c_int, [vp, vp, i64, i64, vp, i32, i32, f32, i32, i32, vp, vp]),
    "ampli_acc_to_slices": (C.c_int, [vp, C.POINTER(AccTable), i32, vp, vp]),
    "ampli_error_finalize_slice": (C.c_int, [vp, i64, i32, i32, vp, vp, f32, i32, vp]),
    "ampli_error_table_unslice": (C.c_int, [vp, i64, i32, vp, vp, vp, vp, vp, vp, vp]),
    "ampli_acc_merge": (C.c_int, [vp, C.POINTER(AccTable), C.POINTER(AccTable), i32]),
    "ampli_acc_packed_len": (i64, [i64]),
    "ampli_acc_pack": (C.c_int, [vp, C.POINTER(AccTable), vp]),
    "ampli_acc_unpack": (C.c_int, [vp, vp, C.POINTER(AccTable)]),
    "ampli_acc_regions": (C.c_int, [i64, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
    "ampli_gm_merge": (C.c_int, [vp, C.POINTER(AccTable), vp, i32]),
    "ampli_error_finalize": (C.c_int, [vp, C.POINTER(AccTable), f32, i32, vp, vp, vp, vp, vp, vp]),
    "ampli_poisson_call": (C.c_int, [vp, vp, i64, i64, vp, i32, vp, vp, i32, i32, vp, vp, i64, vp, vp, vp]),
    "ampli_poisson_call_blocks": (C.c_int, [vp, vp, i64, i64, vp, i32, vp, i32, vp, i32, i32, vp, vp, i64, vp, vp, vp]),
    "ampli_comm_create": (C.c_int, [vp, i32, i32, C.c_char_p, i32, C.POINTER(vp)]),
    "ampli_comm_destroy": (None, [vp]),
    "ampli_comm_reduce_scatter_f64": (C.c_int, [vp, vp, vp, i64]),
    "ampli_comm_all_to_all_f32": (C.c_int, [vp, vp, vp, i64]),
    "ampli_comm_all_gather_bytes": (C.c_int, [vp, vp, vp, i64]),
    "ampli_comm_all_reduce_max_i32": (C.c_int, [vp, C.POINTER(i32), i32]),
    "ampli_comm_exclusive_sum_i64": (C.c_int, [vp, i64, C.POINTER(i64)]),
    "ampli_comm_barrier": (C.c_int, [vp]),
    "ampli_pileup_count": (C.c_int, [vp, vp, vp, i64, vp, i64, i32, i32, vp, vp]),
    "ampli_score_batch": (C.c_int, [vp, vp, vp, vp, i64, vp, vp]),
    "ampli_score_dense_batch": (C.c_int, [vp, vp, vp, vp, i64, vp]),
    "ampli_roundtrip_batch": (C.c_int, [vp, vp, i64, vp]),
    "ampli_synth_fill": (C.c_int, [vp, vp, i64, i32, i32, u64, i32, i32]),
    "ampli_synth_ref": (C.c_int, [vp, vp, i64, u64]),
    "ampli_set_ranges": (C.c_int, [vp, i32]),
    "ampli_ranges_join": (C.c_int, [vp]),
    "ampli_ranges_concurrent": (C.c_int, [vp]),
    "ampli_range_event_record": (C.c_int, [vp, i32, vp]),
    "ampli_set_reduce_compact": (C.c_int, [vp, i32]),
    "ampli_last_reduce_kernel": (C.c_int, [vp]),
    "ampli_set_tuning": (C.c_int, [vp, i32, i32, i32]),
    "ampli_ctx_flags": (C.c_int, [vp, C.POINTER(i32), i32]),
    "ampli_set_queue_items": (C.c_int, [vp, i64]),
    "ampli_set_poisson_tuning": (C.c_int, [vp, i32, i32]),
    "ampli_error_reduce_records": (C.c_int, [vp, C.POINTER(Records), i64, i32, f32, i32, C.POINTER(AccTable), i32, vp, vp, vp, vp, vp, vp]),
    "ampli_error_sums_inorder": (C.c_int, [vp, C.POINTER(Records), i64, f32, i32, C.POINTER(AccTable), i32]),
    "ampli_error_reduce_records_sliced": (C.c_int, [vp, C.POINTER(Records), i64, i32, f32, i32, C.POINTER(AccTable), i32, i32, vp, vp]),
    "ampli_poisson_call_records": (C.c_int, [vp, C.POINTER(Records), i64, vp, vp, i32, i32, vp, vp, i64, vp, vp, vp]),
    "ampli_graph_begin": (C.c_int, [vp]),
    "ampli_graph_end": (C.c_int, [vp, C.POINTER(vp)]),
    "ampli_graph_launch": (C.c_int, [vp, vp]),
    "ampli_graph_destroy": (C.c_int, [vp]),
    "ampli_set_async_drain": (C.c_int, [vp, i32]),
    "ampli_wait_calls": (C.c_int, [vp]),
}

class HostShard(C.Structure):
    """mirror of ampli_host_shard (include/amplisolve_host.h): one shard of a multi-process run + its collective hooks"""
    EE_BUFFERS = C.CFUNCTYPE(C.c_int, vp, i64, C.POINTER(vp))
    HOOK = C.CFUNCTYPE(C.c_int, vp)
    OR_FLAGS = C.CFUNCTYPE(C.c_int, vp, C.POINTER(i32))
    ROWS_BEFORE = C.CFUNCTYPE(C.c_int, vp, i64, C.POINTER(i64))
    _fields_ = [("index", i32), ("count", i32), ("user", vp), ("ee_buffers", EE_BUFFERS), ("ee_exchange", HOOK),
                ("ee_gather", HOOK), ("or_flags", OR_FLAGS), ("rows_before", ROWS_BEFORE), ("barrier", HOOK)]


CHUNK_FN = C.CFUNCTYPE(C.c_int, vp, i32, i32, i32, i64, i64, vp, vp, vp, vp, vp, vp, vp, i64)

HOST_SYMBOLS = {
    "ampli_host_synth_fill": (C.c_int, [vp, i64, i32, i32, u64, i32, i32]),
    "ampli_host_synth_ref": (C.c_int, [vp, i64, u64]),
    "ampli_host_synth_write_panel": (C.c_int, [C.c_char_p, C.c_char_p, i64, u64]),
    "ampli_host_synth_write_aseq": (i64, [C.c_char_p, C.c_char_p, i64, i32, i32, u64, i32, i32, i32]),
    "ampli_host_text_roundtrip_batch": (None, [vp, i64, vp]),
    "ampli_host_af_limit": (i32, [i32]),
    "ampli_host_af_limit_batch": (None, [vp, i64, vp]),
    "ampli_host_af_limit_f32_batch": (None, [vp, i64, vp]),
    "ampli_host_prefilter_nocall": (C.c_int, [i32, i32, f32]),
    "ampli_host_prefilter_skip_f32": (C.c_int, [i32, i32, f32]),
    "ampli_host_drain_score_batch": (None, [vp, vp, vp, i64, vp, vp]),
    "ampli_host_dense_score_batch": (None, [vp, vp, vp, i64, vp]),
    "ampli_host_last_error": (C.c_char_p, []),
    "ampli_host_cohort_load": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]),
    "ampli_host_cohort_load_shard": (C.c_int, [C.c_char_p, C.c_int, C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, i32, i32, C.POINTER(vp)]),
    "ampli_host_cohort_first_sample": (i32, [vp]),
    "ampli_host_cohort_total_samples": (i32, [vp]),
    "ampli_host_cohort_free": (None, [vp]),
    "ampli_host_cohort_P": (i64, [vp]),
    "ampli_host_cohort_E": (i64, [vp]),
    "ampli_host_cohort_S": (i32, [vp]),
    "ampli_host_cohort_walk_len": (i64, [vp]),
    "ampli_host_cohort_recs": (vp, [vp]),
    "ampli_host_cohort_dup_off": (vp, [vp]),
    "ampli_host_cohort_ext_pos": (vp, [vp]),
    "ampli_host_cohort_line_no": (vp, [vp]),
    "ampli_host_cohort_irregular": (vp, [vp, C.POINTER(i64)]),
    "ampli_host_cohort_ref_code": (vp, [vp]),
    "ampli_host_cohort_dup_flag": (vp, [vp]),
    "ampli_host_cohort_sample_name": (C.c_char_p, [vp, i32]),
    "ampli_host_cohort_stats": (None, [vp, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
    "ampli_host_position": (C.c_int, [vp, i64, C.c_char_p, C.c_int, C.POINTER(i32)]),
    "ampli_host_sample_order": (C.c_int, [C.c_char_p, C.c_char_p, i64]),
    "ampli_host_stream_chunks": (C.c_int, [vp, C.c_char_p, C.c_int, C.c_int, i64, CHUNK_FN, vp]),
    "ampli_host_write_error_table": (C.c_int, [vp, vp, vp, vp, vp, C.c_char_p]),
    "ampli_host_read_error_table": (C.c_int, [C.c_char_p, C.POINTER(vp), vp, i64]),
    "ampli_host_read_error_table_vcf": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(vp), vp, i64]),
    "ampli_host_table_cell": (C.c_char_p, [vp, i64, i32]),
    "ampli_host_context": (C.c_int, [vp, i64, C.c_char, C.c_char_p, C.c_char_p, i32]),
    "ampli_host_run_error_estimation": (C.c_int, [C.c_char_p] * 8),
    "ampli_host_run_variant_calling": (C.c_int, [C.c_char_p] * 5),
    "ampli_host_run_error_estimation_sharded": (C.c_int, [C.c_char_p] * 8 + [C.POINTER(HostShard)]),
    "ampli_host_run_variant_calling_sharded": (C.c_int, [C.c_char_p] * 5 + [C.POINTER(HostShard)]),
    "ampli_host_compute_counts": (C.c_int, [C.c_char_p, C.c_char_p, C.c_char_p, i32, i32, i32, i32, C.POINTER(i64)]),
    "ampli_host_bam_scan": (C.c_int, [C.c_char_p, i32, C.POINTER(i64)]),
    "ampli_host_fisher": (C.c_double, [C.c_int] * 4),
    "ampli_host_fisher_direct": (C.c_double, [C.c_int] * 4),
    "ampli_host_guard_score": (C.c_double, [i32, i32, f32, C.POINTER(i32), C.POINTER(i32)]),
}

_hip = None
_host = None


def _bind(lib, table):
    for name, (res, args) in table.items():
        fn = getattr(lib, name)  # AttributeError = missing export: let it propagate
        fn.restype = res
        fn.argtypes = args
    return lib


def hip_lib():
    """Load libamplisolve_hip.so (needs the ROCm runtime, not a GPU, to load)."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB_PATH):
            raise AmpliError(f"{HIP_LIB_PATH} is not built: run `python -m amplisolve_amd.build`; there is no CPU fallback")
        _hip = _bind(C.CDLL(HIP_LIB_PATH), HIP_SYMBOLS)
    return _hip


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise AmpliError(f"{HOST_LIB_PATH} is not built: run `python -m amplisolve_amd.build`")
        _host = _bind(C.CDLL(HOST_LIB_PATH), HOST_SYMBOLS)
    return _host
