// amplisolve_amd/csrc/ampli_internal.h -- what the translation units of libamplisolve_hip.so share: the context behind
// ampli_ctx and the error plumbing.  Not part of the ABI (include/amplisolve_hip.h is).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "../../include/amplisolve_hip.h"

// ---------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------
constexpr int AMPLI_MAX_RANGES = 4;
// poisson_call's prefilter queue (PcItem) + its shard counters (two counter arrays, used alternately)
struct AmpliQueue {
    void *items = nullptr;
    size_t n_items = 0;
    unsigned long long *n = nullptr;
    unsigned parity = 0;
};
// one position range of a context with ranges: its stream, its queue, the event the context's stream waits for when the section closes
struct AmpliLane {
    hipStream_t stream = nullptr; // created by ampli_set_ranges, range 0's too (the context's own stream only forks and joins)
    hipEvent_t done = nullptr;
    bool verified = false; // its stream overlaps with every earlier lane's (checked by ampli_set_ranges)
    AmpliQueue q;
};

struct ampli_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::string err;
    void *ws = nullptr; // workspace for partial accumulator tables
    size_t ws_bytes = 0;
    int reduce_splits = 0; // 0 = auto
    int reduce_groups = 0;  // lane groups per wave in error_reduce: 0 = auto, else 1, 2 or 4
    int reduce_general = 0; // 1 = literal kernel (any depth), 0 = fast kernel (depths < 2^22)
    int last_reduce_kernel = -1; // 0 general, 1 compact uint16, 2 compact 24-bit (ampli_last_reduce_kernel)
    int reduce_compact_u16_only = 0; // ampli_set_reduce_compact(ctx, 2)
    int reduce_compact = 1; // 1 = error_reduce_u16_kernel (compact state, five waves per SIMD) where its shape applies (ampli_set_reduce_compact)
    int grp_size = 1, grp_index = 0; // sliced exchange buffers hold grp_size batches per slice chunk; calls address batch grp_index
    int slice_fmt = 0;               // AMPLI_SLICE_WIDE / AMPLI_SLICE_SLIM: what the sums of the sliced exchange look like (ampli_set_slice_format)
    int rec_layout = 0;     // record layout of every d_recs / d_trecs argument: AMPLI_RECORDS_I32 / _U16 / _U24
    int *d_flags = nullptr; // device word: AMPLI_FLAG_* raised by kernels of this context
    size_t queue_min_items = 0; // ampli_set_queue_items
    // Position ranges on concurrent streams (ampli_set_ranges): every lane owns its stream.  Every lane has its poisson_call prefilter
    // queue (lane 0's is the only one a context without ranges ever uses, on the context's stream).
    int n_ranges = 1;
    AmpliLane lanes[AMPLI_MAX_RANGES];
    hipEvent_t ev_fork = nullptr;
    bool ranges_open = false; // the lanes hold work the context's stream has not waited for yet
    long long ranges_P = 0;   // the panel the open section is cut for
    int ranges_verified = 0;  // 1: every pair of the lanes' streams was seen to run concurrently (ampli_ranges_concurrent)
    // optional: the drain kernel of poisson_call on a side stream (ampli_set_async_drain)
    int async_drain = 0;
    hipStream_t side = nullptr;
    hipEvent_t ev_stream_done = nullptr, ev_drain_done = nullptr;
    bool drain_pending = false;
    int n_cu = 256;
    int sticky = 0; // an error met where nothing could return it (joining the position ranges inside main_stream()): reported by check_launch()
    // poisson_call tuning (ampli_set_poisson_tuning; 0 = default)
    int pc_rows_per_wave = 0, pc_drain_blocks = 0;
    // kf_lgamma at the integers 0 .. AMPLI_LGTAB - 1, filled by the device's own ampli_kf_lgamma (the all-scores mode's scorer)
    double *d_lgtab = nullptr;
};
// 65537 entries (512 KB, L2-resident): every count a uint16 record can hold, + 1 for the drain's kf_lgamma(k + 1).  (4096 until round 6:
// a wave of the drain in which ONE lane carries a count beyond the table -- a heterozygous site at 10 000 x -- runs the Lanczos form, 8
// divisions and 2 logarithms, for all of its lanes.)
constexpr int AMPLI_LGTAB = 65537;

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) {                                                                   \
            if (ctx) (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e_);              \
            return AMPLI_E_HIP;                                                                   \
        }                                                                                         \
    } while (0)

static inline int fail(ampli_ctx *ctx, int code, const char *msg)
{
    if (ctx) ctx->err = msg;
    return code;
}

static inline int check_launch(ampli_ctx *ctx, const char *what)
{
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ctx->err = std::string(what) + ": " + hipGetErrorString(e);
        return AMPLI_E_HIP;
    }
    if (ctx->sticky) { // ctx->err says what it was
        const int rc = ctx->sticky;
        ctx->sticky = 0;
        return rc;
    }
    return AMPLI_OK;
}


// Every ordinary entry point enqueues on main_stream(ctx): if position ranges are still running on their own streams
// (ampli_set_ranges), the context's stream first waits for them -- so whatever follows sees their outputs and may overwrite their
// inputs.  Only the fork / join of the ranges and the capture check touch ctx->stream as it is.
int ampli_ranges_join_internal(ampli_ctx *ctx);
static inline hipStream_t main_stream(ampli_ctx *ctx)
{
    if (ctx->ranges_open) (void)ampli_ranges_join_internal(ctx);
    return ctx->stream;
}
