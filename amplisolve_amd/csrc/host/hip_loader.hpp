// amplisolve_amd/csrc/host/hip_loader.hpp -- run-time binding of libamplisolve_hip.so (include/amplisolve_hip.h).
// The host library must load on a machine without ROCm devices (parsers, writers and their tests run there);
// every compute entry point goes through this table and fails loudly when the library is absent.
#pragma once
#include <string>

#include "../../../include/amplisolve_hip.h"

namespace ampli {

struct HipApi {
    int (*abi_version)(void);
    const char *(*strerror_)(int);
    int (*device_count)(void);
    int (*ctx_create)(int, void *, ampli_ctx **);
    void (*ctx_destroy)(ampli_ctx *);
    const char *(*last_error)(ampli_ctx *);
    int (*sync)(ampli_ctx *);
    int (*pinned_alloc)(size_t, void **);
    int (*pinned_free)(void *);
    int (*host_register)(ampli_ctx *, void *, size_t);
    int (*host_unregister)(void *);
    int (*dev_alloc)(ampli_ctx *, size_t, void **);
    int (*dev_free)(ampli_ctx *, void *);
    int (*copy_h2d)(ampli_ctx *, void *, const void *, size_t);
    int (*copy_d2h)(ampli_ctx *, void *, const void *, size_t);
    int (*memset_d)(ampli_ctx *, void *, int, size_t);
    size_t (*acc_bytes)(int64_t);
    int (*acc_bind)(void *, int64_t, ampli_acc_table *);
    int (*error_reduce)(ampli_ctx *, const int32_t *, int64_t, int64_t, const uint32_t *, int32_t, int32_t, float, int32_t,
                        const ampli_acc_table *);
    int (*error_estimate)(ampli_ctx *, const int32_t *, int64_t, int64_t, const uint32_t *, int32_t, float, int32_t,
                          const ampli_acc_table *, float *, uint8_t *, float *, float *, uint8_t *, int32_t *);
    int (*error_finalize)(ampli_ctx *, const ampli_acc_table *, float, int32_t, float *, uint8_t *, float *, float *, uint8_t *,
                          int32_t *);
    int64_t (*slice_len)(int64_t, int32_t);
    int (*slice_bytes)(int64_t, int32_t, size_t *, size_t *, size_t *);
    int (*error_reduce_sliced)(ampli_ctx *, const int32_t *, int64_t, int64_t, const uint32_t *, int32_t, int32_t, float, int32_t,
                               int32_t, double *, float *);
    int (*error_finalize_slice)(ampli_ctx *, int64_t, int32_t, int32_t, const double *, const float *, float, int32_t, void *);
    int (*error_table_unslice)(ampli_ctx *, int64_t, int32_t, const void *, float *, uint8_t *, float *, float *, uint8_t *, int32_t *);
    int (*set_tuning)(ampli_ctx *, int32_t, int32_t, int32_t);
    int (*ctx_flags)(ampli_ctx *, int32_t *, int32_t);
    int (*set_queue_items)(ampli_ctx *, int64_t);
    int (*poisson_call)(ampli_ctx *, const int32_t *, int64_t, int64_t, const uint32_t *, int32_t, const float *,
                        const uint8_t *, int32_t, int32_t, uint8_t *, ampli_call *, int64_t, unsigned long long *, double *,
                        float *);
    // streamed cohorts: one launch per uploaded chunk of samples
    int (*error_reduce_records)(ampli_ctx *, const ampli_records *, int64_t, int32_t, float, int32_t, const ampli_acc_table *, int32_t, float *,
                                uint8_t *, float *, float *, uint8_t *, int32_t *);
    int (*poisson_call_records)(ampli_ctx *, const ampli_records *, int64_t, const float *, const uint8_t *, int32_t, int32_t, uint8_t *,
                                ampli_call *, int64_t, unsigned long long *, double *, float *);
    int (*acc_to_slices)(ampli_ctx *, const ampli_acc_table *, int32_t, double *, float *);
    int (*error_reduce_records_sliced)(ampli_ctx *, const ampli_records *, int64_t, int32_t, float, int32_t, const ampli_acc_table *, int32_t, int32_t,
                                       double *, float *);
    int (*last_reduce_kernel)(const ampli_ctx *);
    int (*error_sums_inorder)(ampli_ctx *, const ampli_records *, int64_t, float, int32_t, const ampli_acc_table *, int32_t);
    int (*event_create)(void **);
    int (*event_destroy)(void *);
    int (*event_record)(ampli_ctx *, void *);
    int (*event_sync)(void *);
    int (*pileup_count)(ampli_ctx *, const uint8_t *, const uint64_t *, int64_t, const uint64_t *, int64_t, int32_t, int32_t, int32_t *, uint64_t *);
    // native transport of the multi-GPU merge (RCCL over xGMI, include/amplisolve_hip.h "ampli_comm")
    int (*comm_create)(ampli_ctx *, int32_t, int32_t, const char *, int32_t, ampli_comm **);
    void (*comm_destroy)(ampli_comm *);
    int (*comm_reduce_scatter_f64)(ampli_comm *, const double *, double *, int64_t);
    int (*comm_all_to_all_f32)(ampli_comm *, const float *, float *, int64_t);
    int (*comm_all_gather_bytes)(ampli_comm *, const void *, void *, int64_t);
    int (*comm_all_reduce_max_i32)(ampli_comm *, int32_t *, int32_t);
    int (*comm_exclusive_sum_i64)(ampli_comm *, int64_t, int64_t *);
    int (*comm_barrier)(ampli_comm *);
};

const HipApi *hip_api(std::string *why);

} // namespace ampli
