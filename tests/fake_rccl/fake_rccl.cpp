// tests/fake_rccl/fake_rccl.cpp -- TEST DOUBLE, never shipped: the handful of RCCL entry points libamplisolve_hip.so binds
// (ampli_comm_*), implemented over POSIX shared memory so that SEVERAL ranks can share ONE GPU (RCCL itself refuses two ranks
// on a device).  It lets tests/test_gpu_multi_cli.py run the executables' native one-process-per-GPU mode with 2-4 ranks on a
// one-GPU box and check what the real library cannot show there: buffer shapes, counts, chunk order and the ordering of the
// collectives with the kernels.  Every call synchronises the stream, stages through host memory and meets the other ranks at a
// barrier -- slow, blocking, correct.  Selected with AMPLISOLVE_RCCL_LIB=<this .so>.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {
constexpr size_t SLOT = (size_t)96 << 20; // bytes of staging per rank
struct Shared {
    std::atomic<int> arrived;
    std::atomic<int> generation;
    std::atomic<int> attached;
};
struct Comm {
    int rank, world;
    char name[64];
    Shared *sh;
    char *slots; // [world][SLOT]
    size_t bytes;
    struct P2P { bool send; const void *src; void *dst; size_t bytes; int peer; };
    std::vector<P2P> group;
    bool grouping = false;
};
thread_local Comm *g_group_comm = nullptr;

void barrier(Comm *c)
{
    const int gen = c->sh->generation.load();
    if (c->sh->arrived.fetch_add(1) + 1 == c->world) {
        c->sh->arrived.store(0);
        c->sh->generation.fetch_add(1);
    } else {
        for (long spins = 0; c->sh->generation.load() == gen; ++spins) {
            usleep(50);
            if (spins > 20L * 60 * 1000 * 1000 / 50) { fprintf(stderr, "fake_rccl: barrier timed out\n"); abort(); }
        }
    }
}
size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
    }
}
char *slot(Comm *c, int r) { return c->slots + (size_t)r * SLOT; }
bool d2slot(Comm *c, const void *d, size_t n, hipStream_t s, size_t off = 0)
{
    if (off + n > SLOT) { fprintf(stderr, "fake_rccl: message of %zu bytes exceeds the staging slot\n", off + n); return false; }
    return hipStreamSynchronize(s) == hipSuccess && hipMemcpy(slot(c, c->rank) + off, d, n, hipMemcpyDeviceToHost) == hipSuccess;
}
} // namespace

extern "C" {
const char *ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : "fake_rccl error"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    snprintf(id->internal, sizeof id->internal, "/fake_rccl_%d_%ld", (int)getpid(), (long)random());
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int nranks, ncclUniqueId id, int rank)
{
    Comm *c = new Comm();
    c->rank = rank; c->world = nranks;
    snprintf(c->name, sizeof c->name, "%s", id.internal);
    c->bytes = 4096 + (size_t)nranks * SLOT;
    int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { perror("fake_rccl shm"); return ncclSystemError; }
    void *m = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0); // a fresh segment is zero-filled: the counters start at 0
    close(fd);
    if (m == MAP_FAILED) return ncclSystemError;
    c->sh = (Shared *)m;
    c->slots = (char *)m + 4096;
    c->sh->attached.fetch_add(1);
    for (long spins = 0; c->sh->attached.load() < nranks; ++spins) { // everybody has mapped the segment
        usleep(100);
        if (spins > 600000) { fprintf(stderr, "fake_rccl: ranks did not all arrive\n"); return ncclSystemError; }
    }
    barrier(c);
    if (rank == 0) shm_unlink(c->name); // mapped by all: the name can go
    *out = (ncclComm_t)c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    Comm *c = (Comm *)comm;
    munmap((void *)c->sh, c->bytes);
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclReduceScatter(const void *send, void *recv, size_t recvcount, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t s)
{
    Comm *c = (Comm *)comm;
    if (t != ncclFloat64 || op != ncclSum) return ncclInvalidArgument;
    if (!d2slot(c, send, recvcount * 8 * c->world, s)) return ncclSystemError;
    barrier(c);
    std::vector<double> acc(recvcount, 0.0);
    for (int k = 0; k < c->world; ++k) { // rank order, like the ordered merges everywhere else in this code base
        const double *p = (const double *)slot(c, k) + (size_t)c->rank * recvcount;
        for (size_t i = 0; i < recvcount; ++i) acc[i] += p[i];
    }
    if (hipMemcpy(recv, acc.data(), recvcount * 8, hipMemcpyHostToDevice) != hipSuccess) return ncclSystemError;
    barrier(c);
    return ncclSuccess;
}

ncclResult_t ncclAllGather(const void *send, void *recv, size_t sendcount, ncclDataType_t t, ncclComm_t comm, hipStream_t s)
{
    Comm *c = (Comm *)comm;
    const size_t n = sendcount * type_size(t);
    if (!n) return ncclInvalidArgument;
    if (!d2slot(c, send, n, s)) return ncclSystemError;
    barrier(c);
    for (int k = 0; k < c->world; ++k)
        if (hipMemcpy((char *)recv + (size_t)k * n, slot(c, k), n, hipMemcpyHostToDevice) != hipSuccess) return ncclSystemError;
    barrier(c);
    return ncclSuccess;
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, ncclDataType_t t, ncclRedOp_t op, ncclComm_t comm, hipStream_t s)
{
    Comm *c = (Comm *)comm;
    if (t != ncclInt32 || op != ncclMax) return ncclInvalidArgument;
    if (!d2slot(c, send, count * 4, s)) return ncclSystemError;
    barrier(c);
    std::vector<int> acc(count);
    for (size_t i = 0; i < count; ++i) {
        int v = ((const int *)slot(c, 0))[i];
        for (int k = 1; k < c->world; ++k) v = ((const int *)slot(c, k))[i] > v ? ((const int *)slot(c, k))[i] : v;
        acc[i] = v;
    }
    if (hipMemcpy(recv, acc.data(), count * 4, hipMemcpyHostToDevice) != hipSuccess) return ncclSystemError;
    barrier(c);
    return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }

ncclResult_t ncclSend(const void *send, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
    Comm *c = (Comm *)comm;
    g_group_comm = c;
    c->group.push_back({true, send, nullptr, count * type_size(t), peer});
    (void)s;
    return ncclSuccess;
}

ncclResult_t ncclRecv(void *recv, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t s)
{
    Comm *c = (Comm *)comm;
    g_group_comm = c;
    c->group.push_back({false, nullptr, recv, count * type_size(t), peer});
    (void)s;
    return ncclSuccess;
}

// one send and one receive of equal size per peer (what ampli_comm_all_to_all_f32 issues): the message for peer k is staged at
// k * bytes of the sender's slot
ncclResult_t ncclGroupEnd()
{
    Comm *c = g_group_comm;
    if (!c) return ncclSuccess;
    if (hipDeviceSynchronize() != hipSuccess) return ncclSystemError;
    for (auto &p : c->group)
        if (p.send) {
            if ((size_t)(p.peer + 1) * p.bytes > SLOT) return ncclSystemError;
            if (hipMemcpy(slot(c, c->rank) + (size_t)p.peer * p.bytes, p.src, p.bytes, hipMemcpyDeviceToHost) != hipSuccess) return ncclSystemError;
        }
    barrier(c);
    for (auto &p : c->group)
        if (!p.send && hipMemcpy(p.dst, slot(c, p.peer) + (size_t)c->rank * p.bytes, p.bytes, hipMemcpyHostToDevice) != hipSuccess) return ncclSystemError;
    barrier(c);
    c->group.clear();
    g_group_comm = nullptr;
    return ncclSuccess;
}
}
