// amplisolve_amd/csrc/ampli_comm.hip -- ampli_comm_* (include/amplisolve_hip.h): RCCL bound at run time.  No kernels here; the
// file is compiled with the others so that the collectives share the context's stream and error reporting.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h> // types only: librccl.so is dlopen'ed by ampli_comm_create, single-GPU runs never load it

#include <dlfcn.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>

#include "ampli_internal.h"

// ---------------------------------------------------------------------------
// Native transport of the multi-GPU merge: RCCL over xGMI, one process per GPU, without Python.  librccl.so is bound at
// run time (dlopen) the first time a communicator is asked for.  Rendezvous: rank 0 writes the ncclUniqueId, tagged with the
// job it is for (IdFile below), to a file every rank can see (temporary name + rename, so a reader never sees half of it); the
// others poll for it, reject what is not this job's, and nobody waits for a missing rank longer than the timeout.
// Every collective is enqueued on the context's stream, i.e. ordered with the kernels around it.
// ---------------------------------------------------------------------------
struct ampli_comm {
    ampli_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    int *d_small = nullptr; // 64 int64 words of device scratch for the small collectives
};

namespace {
struct RcclApi {
    void *h = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr; // optional
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

bool rccl_load(std::string &why)
{
    if (g_rccl.h) return true;
    const char *cands[] = {getenv("AMPLISOLVE_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char *c : cands) {
        if (!c) continue;
        g_rccl.h = dlopen(c, RTLD_NOW | RTLD_GLOBAL);
        if (g_rccl.h) break;
        why = dlerror();
    }
    if (!g_rccl.h) return false;
#define AMPLI_RCCL_BIND(field, sym)                                             \
    *(void **)(&g_rccl.field) = dlsym(g_rccl.h, sym);                            \
    if (!g_rccl.field) { why = std::string("librccl: missing ") + sym; g_rccl.h = nullptr; return false; }
    AMPLI_RCCL_BIND(GetUniqueId, "ncclGetUniqueId") AMPLI_RCCL_BIND(CommInitRank, "ncclCommInitRank") AMPLI_RCCL_BIND(CommDestroy, "ncclCommDestroy")
    AMPLI_RCCL_BIND(ReduceScatter, "ncclReduceScatter") AMPLI_RCCL_BIND(AllGather, "ncclAllGather") AMPLI_RCCL_BIND(AllReduce, "ncclAllReduce")
    AMPLI_RCCL_BIND(Send, "ncclSend") AMPLI_RCCL_BIND(Recv, "ncclRecv") AMPLI_RCCL_BIND(GroupStart, "ncclGroupStart") AMPLI_RCCL_BIND(GroupEnd, "ncclGroupEnd")
    AMPLI_RCCL_BIND(GetErrorString, "ncclGetErrorString")
#undef AMPLI_RCCL_BIND
    *(void **)(&g_rccl.CommAbort) = dlsym(g_rccl.h, "ncclCommAbort");
    return true;
}
} // namespace

#define RCCL_TRY(c, expr)                                                                                   \
    do {                                                                                                    \
        ncclResult_t r_ = (expr);                                                                           \
        if (r_ != ncclSuccess) {                                                                            \
            (c)->ctx->err = std::string(#expr) + ": " + g_rccl.GetErrorString(r_);                          \
            return AMPLI_E_HIP;                                                                             \
        }                                                                                                   \
    } while (0)

// What rank 0 publishes.  A bare ncclUniqueId (round 2) could not be told from the one a run that died left behind: ranks
// above 0 read the stale id on their first poll and sat in ncclCommInitRank for ever.  Now the file says which job it is for
// and the readers look at its age, and the init itself is bounded.
struct IdFile {
    char magic[8];              // "AMPLRCC2"
    int32_t world;              // size of the job this id was drawn for
    int32_t reserved;
    unsigned long long nonce;   // AMPLISOLVE_JOB_NONCE of the launch (0 = none given)
    long long written_at;       // time(nullptr) at rank 0
    ncclUniqueId id;
};

namespace {
unsigned long long job_nonce()
{
    const char *e = getenv("AMPLISOLVE_JOB_NONCE");
    return e && *e ? strtoull(e, nullptr, 0) : 0ull;
}

// ncclCommInitRank with a deadline.  RCCL has no timeout of its own and a blocking init cannot be cancelled, so it runs on a
// helper thread that owns its state; when the deadline passes the caller gets an error (the process is expected to end with a
// non-zero status -- the helper thread is left behind, detached, and touches nothing of the caller's).
struct InitJob {
    std::mutex m;
    std::condition_variable cv;
    bool done = false;
    ncclResult_t rc = ncclSuccess;
    ncclComm_t comm = nullptr;
};
} // namespace

extern "C" int ampli_comm_create(ampli_ctx *ctx, int32_t rank, int32_t world, const char *id_file, int32_t timeout_s, ampli_comm **out)
{
    if (!ctx || !out || world < 1 || rank < 0 || rank >= world || !id_file || !*id_file) return AMPLI_E_INVALID;
    *out = nullptr;
    if (timeout_s < 1) timeout_s = 1;
    std::string why;
    if (!rccl_load(why)) return fail(ctx, AMPLI_E_HIP, ("librccl.so could not be loaded: " + why).c_str());
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    IdFile rec;
    const unsigned long long nonce = job_nonce();
    const long long entered = (long long)time(nullptr);
    const std::string path(id_file), tmp = path + ".tmp" + std::to_string((long)getpid());
    if (rank == 0) {
        // whatever is there is not ours: a file of this name belongs to a run that did not reach its clean-up
        (void)unlink(path.c_str());
        memset(&rec, 0, sizeof rec);
        memcpy(rec.magic, "AMPLRCC2", 8);
        rec.world = world; rec.nonce = nonce; rec.written_at = entered;
        if (g_rccl.GetUniqueId(&rec.id) != ncclSuccess) return fail(ctx, AMPLI_E_HIP, "ncclGetUniqueId failed");
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(&rec, 1, sizeof rec, f) != sizeof rec) { if (f) fclose(f); return fail(ctx, AMPLI_E_INVALID, ("cannot write " + tmp).c_str()); }
        fclose(f);
        if (rename(tmp.c_str(), path.c_str()) != 0) return fail(ctx, AMPLI_E_INVALID, ("cannot create " + path).c_str());
    } else {
        // accept only a file of THIS job: right magic and world size, the launch's nonce when there is one, and not older than
        // this process could have waited for (a file from before entered - timeout_s is somebody else's)
        bool got = false;
        std::string rejected;
        for (int waited_ms = 0; waited_ms <= timeout_s * 1000 && !got; waited_ms += 20) {
            FILE *f = fopen(path.c_str(), "rb");
            if (f) {
                const bool whole = fread(&rec, 1, sizeof rec, f) == sizeof rec;
                fclose(f);
                if (!whole || memcmp(rec.magic, "AMPLRCC2", 8) != 0) rejected = "not an id file of this library version";
                else if (rec.world != world) rejected = "written for " + std::to_string(rec.world) + " ranks, this job has " + std::to_string(world);
                else if (rec.nonce != nonce) rejected = "AMPLISOLVE_JOB_NONCE differs: the file belongs to another launch";
                else if (rec.written_at < entered - (long long)timeout_s - 5) rejected = "older than this job (left behind by a run that died?)";
                else got = true;
            }
            if (!got) usleep(20000);
        }
        if (got) {
            // the file may be a dead run's that rank 0 is about to replace (it unlinks and rewrites on entry): look once more
            // a moment later and take the newer record
            usleep(250000);
            IdFile again;
            FILE *f = fopen(path.c_str(), "rb");
            if (f) {
                const bool whole = fread(&again, 1, sizeof again, f) == sizeof again;
                fclose(f);
                if (whole && memcmp(again.magic, "AMPLRCC2", 8) == 0 && again.world == world && again.nonce == nonce && again.written_at >= rec.written_at) rec = again;
            }
        }
        if (!got)
            return fail(ctx, AMPLI_E_HIP, ("timed out waiting for rank 0's id file " + path + (rejected.empty() ? "" : " (a file is there but was rejected: " + rejected + ")")).c_str());
    }
    ampli_comm *c = new (std::nothrow) ampli_comm();
    if (!c) return AMPLI_E_NOMEM;
    c->ctx = ctx; c->rank = rank; c->world = world;
    {
        auto job = std::make_shared<InitJob>();
        const ncclUniqueId id = rec.id;
        const int dev = ctx->device;
        std::thread([job, id, world, rank, dev]() {
            (void)hipSetDevice(dev);
            ncclComm_t cm = nullptr;
            const ncclResult_t r = g_rccl.CommInitRank(&cm, world, id, rank);
            std::lock_guard<std::mutex> lk(job->m);
            job->rc = r; job->comm = cm; job->done = true;
            job->cv.notify_all();
        }).detach();
        std::unique_lock<std::mutex> lk(job->m);
        if (!job->cv.wait_for(lk, std::chrono::seconds(timeout_s), [&] { return job->done; })) {
            delete c;
            if (rank == 0) (void)unlink(path.c_str());
            return fail(ctx, AMPLI_E_COMM_TIMEOUT, ("ncclCommInitRank did not complete within " + std::to_string(timeout_s) + " s (AMPLISOLVE_RCCL_TIMEOUT): a rank is missing, or the id file " + path +
                                           " is not this job's; the process should end now").c_str());
        }
        if (job->rc != ncclSuccess) {
            if (job->comm && g_rccl.CommAbort) (void)g_rccl.CommAbort(job->comm);
            delete c;
            return fail(ctx, AMPLI_E_HIP, (std::string("ncclCommInitRank failed: ") + g_rccl.GetErrorString(job->rc)).c_str());
        }
        c->comm = job->comm;
    }
    if (hipMalloc((void **)&c->d_small, 64 * sizeof(long long)) != hipSuccess) { g_rccl.CommDestroy(c->comm); delete c; return AMPLI_E_NOMEM; }
    *out = c;
    return AMPLI_OK;
}

extern "C" void ampli_comm_destroy(ampli_comm *c)
{
    if (!c) return;
    (void)hipStreamSynchronize(main_stream(c->ctx));
    if (c->d_small) (void)hipFree(c->d_small);
    if (c->comm) g_rccl.CommDestroy(c->comm);
    delete c;
}

// sums [world][count] f64 -> this rank's [count] (SUM); xGMI: each rank receives (world-1)/world of count*8 bytes
extern "C" int ampli_comm_reduce_scatter_f64(ampli_comm *c, const double *d_send, double *d_recv, int64_t count)
{
    if (!c || !d_send || !d_recv || count <= 0) return AMPLI_E_INVALID;
    RCCL_TRY(c, g_rccl.ReduceScatter(d_send, d_recv, (size_t)count, ncclFloat64, ncclSum, c->comm, main_stream(c->ctx)));
    return AMPLI_OK;
}

// send [world][count] f32 (chunk k for rank k) -> recv [world][count] (chunk k from rank k): grouped send / recv pairs
extern "C" int ampli_comm_all_to_all_f32(ampli_comm *c, const float *d_send, float *d_recv, int64_t count)
{
    if (!c || !d_send || !d_recv || count <= 0) return AMPLI_E_INVALID;
    RCCL_TRY(c, g_rccl.GroupStart());
    for (int k = 0; k < c->world; ++k) {
        RCCL_TRY(c, g_rccl.Send(d_send + (size_t)k * count, (size_t)count, ncclFloat32, k, c->comm, main_stream(c->ctx)));
        RCCL_TRY(c, g_rccl.Recv(d_recv + (size_t)k * count, (size_t)count, ncclFloat32, k, c->comm, main_stream(c->ctx)));
    }
    RCCL_TRY(c, g_rccl.GroupEnd());
    return AMPLI_OK;
}

extern "C" int ampli_comm_all_gather_bytes(ampli_comm *c, const void *d_send, void *d_recv, int64_t bytes)
{
    if (!c || !d_send || !d_recv || bytes <= 0) return AMPLI_E_INVALID;
    RCCL_TRY(c, g_rccl.AllGather(d_send, d_recv, (size_t)bytes, ncclUint8, c->comm, main_stream(c->ctx)));
    return AMPLI_OK;
}

// host values, in place: element-wise MAX over the ranks (flags travel as one 0/1 word per bit); synchronises
extern "C" int ampli_comm_all_reduce_max_i32(ampli_comm *c, int32_t *values, int32_t n)
{
    if (!c || !values || n < 1 || n > 64) return AMPLI_E_INVALID;
    HIP_TRY(c->ctx, hipMemcpyAsync(c->d_small, values, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, main_stream(c->ctx)));
    RCCL_TRY(c, g_rccl.AllReduce(c->d_small, c->d_small, (size_t)n, ncclInt32, ncclMax, c->comm, main_stream(c->ctx)));
    HIP_TRY(c->ctx, hipMemcpyAsync(values, c->d_small, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, main_stream(c->ctx)));
    HIP_TRY(c->ctx, hipStreamSynchronize(main_stream(c->ctx)));
    return AMPLI_OK;
}

// sum of `mine` over the ranks below this one (all-gather of one int64 per rank); synchronises
extern "C" int ampli_comm_exclusive_sum_i64(ampli_comm *c, int64_t mine, int64_t *before)
{
    if (!c || !before || c->world > 63) return AMPLI_E_INVALID;
    long long *d = (long long *)c->d_small; // word 0: mine; words 1..world: gathered
    HIP_TRY(c->ctx, hipMemcpyAsync(d, &mine, sizeof(long long), hipMemcpyHostToDevice, main_stream(c->ctx)));
    RCCL_TRY(c, g_rccl.AllGather(d, d + 1, 1, ncclInt64, c->comm, main_stream(c->ctx)));
    long long all[64];
    HIP_TRY(c->ctx, hipMemcpyAsync(all, d + 1, (size_t)c->world * sizeof(long long), hipMemcpyDeviceToHost, main_stream(c->ctx)));
    HIP_TRY(c->ctx, hipStreamSynchronize(main_stream(c->ctx)));
    long long s_ = 0;
    for (int k = 0; k < c->rank; ++k) s_ += all[k];
    *before = s_;
    return AMPLI_OK;
}

extern "C" int ampli_comm_barrier(ampli_comm *c)
{
    int32_t one = 1;
    return ampli_comm_all_reduce_max_i32(c, &one, 1);
}
