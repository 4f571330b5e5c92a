// amplisolve_amd/csrc/host/pipeline.cpp -- the two command lines, end to end.
//   run_error_estimation  re-states main() of AmpliSolveErrorEstimation.cpp (EE:241-520)
//   run_variant_calling   re-states main() + callVariants of AmpliSolveVariantCalling.cpp (VC:199-360, VC:633-3304)
// Parsing / formatting happen here; sums, rates, p-values and the call gate come from libamplisolve_hip.so.
#include <sys/stat.h>
#include <sys/types.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <iomanip>
#include <iostream>
#include <memory>
#include <mutex>
#include <sstream>
#include <thread>

#include "hip_loader.hpp"
#include "host.hpp"

namespace ampli {

// ---- PhaseClock ----
namespace {
struct PhaseEntry { std::string name; double s; bool critical; };
std::mutex g_phase_mu;
std::vector<PhaseEntry> g_phases;
} // namespace
double PhaseClock::now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
void PhaseClock::add(const char *name, double seconds, bool critical)
{
    std::lock_guard<std::mutex> lk(g_phase_mu);
    for (auto &e : g_phases)
        if (e.name == name && e.critical == critical) { e.s += seconds; return; }
    g_phases.push_back(PhaseEntry{name, seconds, critical});
}
void PhaseClock::reset()
{
    std::lock_guard<std::mutex> lk(g_phase_mu);
    g_phases.clear();
}
void PhaseClock::report(std::ostream &os, double wall)
{
    std::lock_guard<std::mutex> lk(g_phase_mu);
    double sum = 0;
    for (auto &e : g_phases) {
        os << "TIMING2 " << e.name << " " << e.s << (e.critical ? " critical" : " overlapped") << "\n";
        if (e.critical) sum += e.s;
    }
    // the wall clock (CLOCK_REALTIME) at this report, for a parent that wants to split what lies outside main() into the time
    // before main was entered and the time after the report (process teardown)
    const double epoch = std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
    os << "TIMING2 unattributed " << wall - sum << " critical\nTIMING2 wall_in_main " << wall << " total\nTIMING2 epoch_at_report "
       << std::fixed << std::setprecision(6) << epoch << " total" << std::endl;
    os.unsetf(std::ios::fixed);
}

namespace {

void mkdir_p(const std::string &path) // generateFolder: `mkdir -p` (EE:3079-3086)
{
    std::string cur;
    for (size_t i = 0; i <= path.size(); ++i) {
        if (i == path.size() || path[i] == '/') {
            if (!cur.empty()) mkdir(cur.c_str(), 0777);
        }
        if (i < path.size()) cur.push_back(path[i]);
    }
}

double now_s() { return PhaseClock::now(); }

struct Dev {
    const HipApi *api = nullptr;
    ampli_ctx *ctx = nullptr;
    std::vector<void *> allocs;
    ~Dev()
    {
        if (ctx) {
            PhaseClock::Scope sc("device_teardown");
            for (void *p : allocs) api->dev_free(ctx, p);
            api->ctx_destroy(ctx);
        }
    }
    void check(int rc, const char *what)
    {
        if (rc != AMPLI_OK)
            throw Error{rc, std::string(what) + ": " + api->strerror_(rc) + (ctx ? std::string(" -- ") + api->last_error(ctx) : "")};
    }
    bool side = false; // opened on a side thread: its start-up spans are overlapped work, not the main thread's path
    void open()
    {
        std::string why;
        {
            PhaseClock::Scope sc("hip_library_load", !side); // dlopen of libamplisolve_hip.so: pulls in the HIP runtime, registers the code object
            api = hip_api(&why);
        }
        if (!api) throw Error{AMPLI_E_HIP, "libamplisolve_hip.so could not be loaded (" + why + "); there is no CPU fallback"};
        {
            PhaseClock::Scope sc("runtime_init", !side); // the first HIP call of the process: HSA / driver start-up
            if (api->device_count() <= 0) throw Error{AMPLI_E_HIP, "no MI355X visible; there is no CPU fallback"};
        }
        int dev = 0;
        if (const char *e = getenv("AMPLISOLVE_DEVICE")) dev = atoi(e);
        {
            PhaseClock::Scope sc("context_create", !side); // hipSetDevice + properties + the first hipMalloc / hipMemset
            check(api->ctx_create(dev, nullptr, &ctx), "ampli_ctx_create");
        }
        if (side) warm_copies();
    }
    // The first copy in each direction sets up the runtime's copy machinery (staging buffers, the DMA queues: ~8 ms each,
    // tools/micro/init_probe.cpp).  On the side thread that cost hides behind the parsers; paid later it sits on the main
    // thread's path, in front of the first upload and of the table download.
    void warm_copies()
    {
        PhaseClock::Scope sc("warm_copies", false);
        const size_t n = 1 << 16;
        void *d = nullptr, *pin = nullptr;
        if (api->dev_alloc(ctx, n, &d) != AMPLI_OK) return;
        std::vector<char> pageable(n, 1);
        if (api->pinned_alloc(n, &pin) == AMPLI_OK) { // the uploads come from pinned (registered) memory
            memset(pin, 1, n);
            (void)api->copy_h2d(ctx, d, pin, n);
            (void)api->copy_d2h(ctx, pin, d, n);
        }
        (void)api->copy_h2d(ctx, d, pageable.data(), n); // small host arrays and the results travel pageable
        (void)api->copy_d2h(ctx, pageable.data(), d, n);
        (void)api->sync(ctx);
        if (pin) (void)api->pinned_free(pin);
        (void)api->dev_free(ctx, d);
    }
    template <class T> T *alloc(size_t n)
    {
        PhaseClock::Scope sc("device_alloc");
        void *p = nullptr;
        check(api->dev_alloc(ctx, n * sizeof(T), &p), "ampli_dev_alloc");
        allocs.push_back(p);
        return (T *)p;
    }
    void free(void *p)
    {
        PhaseClock::Scope sc("device_alloc");
        for (auto it = allocs.begin(); it != allocs.end(); ++it)
            if (*it == p) { allocs.erase(it); break; }
        check(api->dev_free(ctx, p), "ampli_dev_free");
    }
    void h2d(void *d, const void *src, size_t bytes)
    {
        PhaseClock::Scope sc("h2d_enqueue");
        check(api->copy_h2d(ctx, d, src, bytes), "ampli_copy_h2d");
    }
    template <class T> T *upload(const T *src, size_t n)
    {
        T *d = alloc<T>(n ? n : 1);
        if (n) h2d(d, src, n * sizeof(T));
        return d;
    }
    template <class T> void download(T *dst, const T *d, size_t n)
    {
        PhaseClock::Scope sc("d2h");
        check(api->copy_d2h(ctx, dst, d, n * sizeof(T)), "ampli_copy_d2h");
    }
    void sync()
    {
        PhaseClock::Scope sc("device_wait");
        check(api->sync(ctx), "ampli_sync");
    }
};

// a device buffer that only ever grows (one per ring slot and kind)
struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    void *ensure(Dev &dev, size_t bytes)
    {
        if (bytes > cap) {
            if (p) dev.free(p);
            p = nullptr;
            cap = bytes + bytes / 8 + 256;
            p = dev.alloc<char>(cap);
        }
        return p;
    }
};

struct DevSlot {
    DevBuf prim, ext, aux, mask, rd, rd_ext;
};

// The context is opened on a side thread while the main thread reads the panel / the error table: loading the HIP runtime and
// the code object takes 0.1-0.2 s, a good part of a command line's wall time on small and medium cohorts.
struct DevAsync {
    Dev dev;
    std::thread th;
    std::exception_ptr ex;
    bool started = false;
    void start()
    {
        started = true;
        dev.side = true;
        th = std::thread([this] {
            try {
                dev.open();
            } catch (...) {
                ex = std::current_exception();
            }
        });
    }
    Dev &get() // the opened context; rethrows what open() threw (no device, no library: there is no CPU fallback)
    {
        if (!started) start();
        {
            PhaseClock::Scope sc("wait_for_context"); // what of the start-up the panel / table parsing did not hide
            if (th.joinable()) th.join();
        }
        if (ex) std::rethrow_exception(ex);
        return dev;
    }
    ~DevAsync()
    {
        if (th.joinable()) th.join();
    }
};

// work that nothing downstream waits for (by-product files, freeing the ring): runs beside the main thread, joined when the
// owner leaves its scope; an exception is rethrown by wait()
struct Background {
    std::thread th;
    std::exception_ptr ex;
    template <class F> void run(F &&f)
    {
        wait();
        th = std::thread([this, f]() mutable {
            try {
                f();
            } catch (...) {
                ex = std::current_exception();
            }
        });
    }
    void wait()
    {
        if (th.joinable()) th.join();
        if (ex) {
            std::exception_ptr e = ex;
            ex = nullptr;
            std::rethrow_exception(e);
        }
    }
    ~Background()
    {
        if (th.joinable()) th.join();
    }
};

// The ring is up to 512 MB of touched, pinned memory.  Freeing it on a background thread takes ~35 ms during which the address
// space is write-locked again and again: the table writer / the annotation threads beside it stall in their own page faults
// (write_table 8 -> 30 ms on config 3).  An executable that is about to _exit leaves the buffers to the exit instead
// (AMPLISOLVE_RING_TEARDOWN=background restores the freeing, for comparison).
bool leave_ring_to_exit(const bool process_ends)
{
    if (const char *e = getenv("AMPLISOLVE_RING_TEARDOWN")) return std::string(e) == "exit";
    return process_ends;
}

size_t chunk_bytes_setting()
{
    // about this many bytes of records (in the narrowest layout) per chunk (AMPLISOLVE_CHUNK_MB); ring_slots_setting() chunks may be parsed ahead
    size_t mb = 128;
    if (const char *e = getenv("AMPLISOLVE_CHUNK_MB")) mb = (size_t)std::max(1, atoi(e));
    if (const char *e = getenv("AMPLISOLVE_CHUNK_BYTES")) return (size_t)std::max(1ll, atoll(e)); // tests: down to one sample per chunk
    return mb << 20;
}

// Host ring slots.  The consumer needs the device context before it can take a chunk, and the HIP runtime's start-up takes 0.06-0.25 s:
// with the four slots of rounds 2-4 the parsers of a many-chunk cohort filled them and then stood still until the context was up
// (config-4-sized cohort, 13 chunks: wall = start-up + the rest of the parsing instead of the larger of the two).  So the ring may
// hold up to AMPLISOLVE_RING_MB (default 2048) of records in the narrowest layout, at least 4 and at most 64 slots; a slot's memory
// is only touched when a chunk is packed into it, and a cohort of fewer chunks allocates fewer slots.  The device side keeps four
// buffers: a chunk is uploaded, consumed and waited for before the next one is taken, so slot k simply uses device buffer k mod 4.
int ring_slots_setting(const size_t chunk_bytes)
{
    size_t mb = 2048;
    if (const char *e = getenv("AMPLISOLVE_RING_MB")) mb = (size_t)std::max(1, atoi(e));
    const size_t n = (mb << 20) / std::max<size_t>(1, chunk_bytes);
    return (int)std::min<size_t>(64, std::max<size_t>(4, n));
}
constexpr int kDevSlots = 4;

// AMPLISOLVE_PIN: how a ring buffer reaches the device.  "register" (default): the parsers fill plain page-aligned memory
// -- they start before the HIP runtime is up -- and the buffer is pinned (hipHostRegister, ~6 ms per 128 MB) the first
// time it is uploaded from; "none": never pinned, the runtime stages the copy.  (Round 3 allocated the ring with
// hipHostMalloc: 23 ms per 128 MB up front, after the context was up, and 15 ms per 128 MB to free -- tools/micro/init_probe.cpp.)
bool pin_late()
{
    static const bool v = [] {
        const char *e = getenv("AMPLISOLVE_PIN");
        return !(e && std::string(e) == "none");
    }();
    return v;
}

// upload one chunk into its ring slot and describe it for the kernels
ampli_records upload_chunk(Dev &dev, DevSlot &ds, Chunk &c, bool for_calling)
{
    if (pin_late()) c.pin(dev.ctx);
    const size_t rb = record_bytes(c.layout);
    const size_t pb = (size_t)c.n * (size_t)c.P * rb, eb = (size_t)c.n * (size_t)c.E * rb;
    void *d_prim = ds.prim.ensure(dev, pb);
    dev.h2d(d_prim, c.prim, pb);
    ampli_records r;
    memset(&r, 0, sizeof r);
    r.recs = d_prim;
    r.row_stride = c.P;
    r.layout = c.layout;
    r.n_samples = c.n;
    r.E = c.E;
    if (c.E > 0) {
        void *d_ext = ds.ext.ensure(dev, eb);
        dev.h2d(d_ext, c.ext, eb);
        r.ext = d_ext;
        r.ext_stride = c.E;
        const std::vector<uint32_t> &aux = for_calling ? c.ext_pos : c.dup_off;
        void *d_aux = ds.aux.ensure(dev, aux.size() * sizeof(uint32_t));
        dev.h2d(d_aux, aux.data(), aux.size() * sizeof(uint32_t));
        if (for_calling) r.ext_pos = (const uint32_t *)d_aux;
        else r.dup_off = (const uint32_t *)d_aux;
    }
    if (!c.irregular.empty()) {
        // lines whose RD column is not A+C+G+T (EE:1178-1181, VC:762-765): the column travels as an int32 plane beside the
        // records (AMPLI_ABSENT = regular line) and the kernels use it where the reference does (EE:1229, VC:814, VC:895)
        std::vector<int32_t> rd((size_t)c.n * c.P, AMPLI_ABSENT), rde((size_t)c.n * c.E, AMPLI_ABSENT);
        for (const Irregular &x : c.irregular) {
            if ((int64_t)x.record < c.P) rd[(size_t)x.sample * c.P + x.record] = x.rd;
            else rde[(size_t)x.sample * c.E + (x.record - c.P)] = x.rd;
        }
        void *d_rd = ds.rd.ensure(dev, rd.size() * sizeof(int32_t));
        dev.h2d(d_rd, rd.data(), rd.size() * sizeof(int32_t));
        r.rd = (const int32_t *)d_rd;
        if (c.E > 0) {
            void *d_rde = ds.rd_ext.ensure(dev, rde.size() * sizeof(int32_t));
            dev.h2d(d_rde, rde.data(), rde.size() * sizeof(int32_t));
            r.rd_ext = (const int32_t *)d_rde;
        }
        dev.sync(); // the host vectors go out of scope
    }
    return r;
}

// The multi-GPU mode of the executables themselves (one process per GPU, no Python): the exchange steps of
// ampli_host_shard over the HIP library's own RCCL transport (ampli_comm_*).  Owns a context on the device's default
// stream -- the stream the pipeline's kernels run on -- so every collective is ordered with the kernels around it.
struct NativeShard {
    Dev dev;
    ampli_comm *comm = nullptr;
    ampli_host_shard hooks;
    NativeDist nd;
    std::string err;
    bool active = false;

    NativeShard() { memset(&hooks, 0, sizeof hooks); }
    ~NativeShard()
    {
        if (comm) dev.api->comm_destroy(comm);
    }
    void describe(const NativeDist &d)
    {
        nd = d;
        active = d.world > 1 || getenv("AMPLISOLVE_FORCE_NATIVE_DIST") != nullptr; // forced: the RCCL path on a communicator of one
        hooks.index = d.rank; hooks.count = d.world; hooks.user = this;
        hooks.ee_buffers = &NativeShard::s_buffers; hooks.ee_exchange = &NativeShard::s_exchange; hooks.ee_gather = &NativeShard::s_gather;
        hooks.or_flags = &NativeShard::s_or_flags; hooks.rows_before = &NativeShard::s_rows_before; hooks.barrier = &NativeShard::s_barrier;
    }
    // every rank gets here before it parses anything, so the rendezvous does not wait for the slowest parser
    void open()
    {
        std::string why;
        const HipApi *api = hip_api(&why);
        if (!api) throw Error{AMPLI_E_HIP, "libamplisolve_hip.so could not be loaded (" + why + "); there is no CPU fallback"};
        if (!getenv("AMPLISOLVE_DEVICE")) { // one GPU per process: rank k takes device k (mod the visible ones)
            const int n = api->device_count();
            if (n <= 0) throw Error{AMPLI_E_HIP, "no MI355X visible; there is no CPU fallback"};
            setenv("AMPLISOLVE_DEVICE", std::to_string(nd.rank % n).c_str(), 1);
        }
        if (nd.id_file.empty()) throw Error{AMPLI_E_INVALID, "multi-GPU run: AMPLISOLVE_ID_FILE (a path every process can see) is not set"};
        dev.open();
        const int crc = dev.api->comm_create(dev.ctx, nd.rank, nd.world, nd.id_file.c_str(), nd.timeout_s, &comm);
        if (crc == AMPLI_E_COMM_TIMEOUT) {
            // a detached helper thread is still inside ncclCommInitRank on this device: unwinding through ~Dev (hipFree,
            // context teardown) or exit()'s static destructors beside it can hang or crash.  Say why and leave at once.
            std::cout << "\t\t\nSomething went wrong: ampli_comm_create: " << dev.api->last_error(dev.ctx)
                      << "\n                                        Sorry but Amplisolve cannot continue..." << std::endl;
            std::cout.flush();
            std::cerr.flush();
            fflush(nullptr);
            _exit(1);
        }
        dev.check(crc, "ampli_comm_create");
        dev.check(dev.api->comm_barrier(comm), "ampli_comm_barrier"); // every rank has read the id
        if (nd.rank == 0) std::remove(nd.id_file.c_str());
    }

    size_t sums_b = 0, gm_b = 0, block_b = 0;
    int64_t L = 0;
    void *bufs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};

    template <class F> static int guarded(void *user, F &&f)
    {
        NativeShard *s = (NativeShard *)user;
        try {
            f(*s);
            return 0;
        } catch (const Error &e) {
            s->err = e.msg;
            return e.code ? e.code : -1;
        }
    }
    static int s_buffers(void *user, int64_t P, void **out)
    {
        return guarded(user, [&](NativeShard &s) {
            const int n = s.nd.world;
            s.L = s.dev.api->slice_len(P, n);
            s.dev.check(s.dev.api->slice_bytes(P, n, &s.sums_b, &s.gm_b, &s.block_b), "ampli_slice_bytes");
            const size_t sizes[6] = {s.sums_b, s.gm_b, s.sums_b / (size_t)n, s.gm_b, s.block_b, s.block_b * (size_t)n};
            for (int i = 0; i < 6; ++i) {
                s.bufs[i] = s.dev.alloc<char>(sizes[i]);
                s.dev.check(s.dev.api->memset_d(s.dev.ctx, s.bufs[i], 0, sizes[i]), "ampli_memset_d"); // padding positions are never written
                out[i] = s.bufs[i];
            }
        });
    }
    static int s_exchange(void *user)
    {
        return guarded(user, [&](NativeShard &s) {
            s.dev.check(s.dev.api->comm_reduce_scatter_f64(s.comm, (const double *)s.bufs[0], (double *)s.bufs[2], 21 * s.L), "ampli_comm_reduce_scatter_f64");
            s.dev.check(s.dev.api->comm_all_to_all_f32(s.comm, (const float *)s.bufs[1], (float *)s.bufs[3], 8 * s.L), "ampli_comm_all_to_all_f32");
        });
    }
    static int s_gather(void *user)
    {
        return guarded(user, [&](NativeShard &s) {
            s.dev.check(s.dev.api->comm_all_gather_bytes(s.comm, s.bufs[4], s.bufs[5], (int64_t)s.block_b), "ampli_comm_all_gather_bytes");
        });
    }
    static int s_or_flags(void *user, int32_t *flags)
    {
        return guarded(user, [&](NativeShard &s) {
            int32_t bits[31];
            for (int b = 0; b < 31; ++b) bits[b] = (*flags >> b) & 1;
            s.dev.check(s.dev.api->comm_all_reduce_max_i32(s.comm, bits, 31), "ampli_comm_all_reduce_max_i32");
            int32_t v = 0;
            for (int b = 0; b < 31; ++b) v |= bits[b] ? (1 << b) : 0;
            *flags = v;
        });
    }
    static int s_rows_before(void *user, int64_t mine, int64_t *before)
    {
        return guarded(user, [&](NativeShard &s) { s.dev.check(s.dev.api->comm_exclusive_sum_i64(s.comm, mine, before), "ampli_comm_exclusive_sum_i64"); });
    }
    static int s_barrier(void *user)
    {
        return guarded(user, [&](NativeShard &s) { s.dev.check(s.dev.api->comm_barrier(s.comm), "ampli_comm_barrier"); });
    }
};

const char *kLine = "************************************************************************************************************************************";

} // namespace

void finish_process(int status)
{
    const char *e = getenv("AMPLISOLVE_EXIT");
    if (e && std::string(e) == "orderly") return;
    std::cout.flush();
    std::cerr.flush();
    fflush(nullptr);
    _exit(status);
}

NativeDist native_dist_from_env(const std::string &output_dir)
{
    NativeDist d;
    if (const char *e = getenv("AMPLISOLVE_WORLD_SIZE")) d.world = std::max(1, atoi(e));
    if (const char *e = getenv("AMPLISOLVE_RANK")) d.rank = atoi(e);
    if (d.rank < 0 || d.rank >= d.world) { d.rank = 0; d.world = 1; }
    if (const char *e = getenv("AMPLISOLVE_ID_FILE")) d.id_file = e;
    else if (!output_dir.empty()) d.id_file = output_dir + "/.amplisolve_rccl_id"; // output_dir is shared by all processes anyway
    if (const char *e = getenv("AMPLISOLVE_RCCL_TIMEOUT")) d.timeout_s = std::max(1, atoi(e));
    return d;
}

// ---------------------------------------------------------------------------------------------------------
int run_error_estimation(const EeArgs &a)
{
    try {
        // EE:328-388: numeric conversion and defaults
        float C_value = (float)std::atof(a.C_value.c_str());
        int cov = std::atoi(a.coverage_cutoff.c_str());
        const bool no_germlines = a.germline_dir == "not_available";
        float default_error = 0.01f;
        std::cout << kLine << "\n" << std::endl;
        std::cout << "                                Error estimation required for AmpliSolveVariantCalling program \n" << std::endl;
        std::cout << "                        MI355X-native build (amplisolve_amd); command line and files as AmpliSolveErrorEstimation\n" << std::endl;
        std::cout << "Execution started under the following parameters:" << std::endl;
        std::cout << "\t1. Panel design                                   : " << a.panel_design << std::endl;
        std::cout << "\t2. Reference genome                               : " << a.reference_genome << std::endl;
        if (no_germlines) {
            default_error = (float)std::atof(a.default_error.c_str());
            if (default_error > 0) {
                std::cout << "\t3. Germline count dir                             : NO germline count files available. Estimation of error is based on platform-specific error level given by user equal to " << default_error << std::endl;
            } else {
                default_error = 0.01f;
                std::cout << "\t3. Germline count dir                             : NO germline count files available. User gave wrong platform-specific error level and the estimation will be based on Error=" << default_error << std::endl;
            }
        } else {
            std::cout << "\t3. Germline count dir                             : " << a.germline_dir << std::endl;
        }
        if (C_value <= 0) {
            C_value = 0.002f;
            std::cout << "\t4. C value                                         : User gave: " << a.C_value << ". The value is converted to 0.002" << std::endl;
        } else {
            std::cout << "\t4. C value                                        : " << C_value << std::endl;
        }
        if (cov <= 0) {
            cov = 100;
            std::cout << "\t5. Coverage cutoff                                  : User gave: " << a.coverage_cutoff << ". The value is converted 100" << std::endl;
        } else {
            std::cout << "\t5. Coverage cutoff                                : " << cov << std::endl;
        }
        std::cout << "\t6. Output dir                                     : " << a.output_dir << std::endl;

        const ampli_host_shard *sh = (a.shard && a.shard->count > 1) ? a.shard : nullptr;
        NativeShard native; // the executables' own multi-GPU mode (RCCL); callers with their own transport pass a.shard
        if (!sh) {
            native.describe(a.native);
            if (native.active) {
                mkdir_p(a.output_dir); // the default id file lives there
                native.open();
                sh = &native.hooks;
            }
        }
        const bool writer = !sh || sh->index == 0; // shard 0 writes every file of a multi-process run
        auto hook = [&](int rc, const char *what) {
            if (rc != 0) throw Error{AMPLI_E_INVALID, std::string("shard hook failed: ") + what + (native.err.empty() ? "" : " -- " + native.err)};
        };
        DevAsync dev_async;
        if (!no_germlines) dev_async.start(); // after the native shard (it may pick the device), beside the panel parsing
        const std::string interm = a.output_dir + "/AmpliSolveErrorEstimation_interm_files"; // EE:414
        if (writer) mkdir_p(interm);
        srand((unsigned)time(nullptr));
        const int seed = rand() % 1000; // EE:581-584

        double t0 = now_s();
        Panel panel;
        Background interm_files, ring_teardown; // declared after the panel: they are joined before it goes away
        {
            PhaseClock::Scope sc("panel");
            panel_from_bed(a.panel_design, panel);
            if (!a.refbases_file.empty()) panel_load_refbases_file(panel, a.refbases_file);
            else panel_load_fasta(panel, a.reference_genome);
            // the five by-product files of generateReferenceBases (EE:601-664) are read by nothing downstream
            if (writer) interm_files.run([&panel, interm, seed] {
                PhaseClock::Scope sc2("interm_files", false);
                panel_write_interm_files(panel, interm, seed);
            });
        }
        std::cout << "\nRunning function generateReferenceBases: Reference bases and amplicon duplicated positions have generated"
                  << "\n\t\t --> Parsed in total " << panel.rows.size() << " amplicons and annotated " << panel.walk.size() << " positions." << std::endl;
        std::cout << "Running function storeReference: panel reference bases stored with success " << panel.P() << std::endl;
        size_t ndup = 0;
        for (auto d : panel.dup) ndup += d;
        std::cout << "Running function storeDuplicates: panel duplicate positions stored with success " << ndup << std::endl;

        if (no_germlines) { // EE:472-506
            const std::string out = a.output_dir + "/positionSpecificNoise_default.txt";
            if (writer) write_error_table_default(panel, default_error, out);
            std::cout << "\nAmpliSolveErrorEstimation execution was successful. Results can be found at: " << out << std::endl;
            std::cout << "\n" << kLine << std::endl;
            return 0;
        }

        double t1 = now_s();
        const std::string list_name = interm + "/" + std::to_string(seed) + "_germline_count_list_original.txt"; // EE:442
        int threads = 0;
        if (const char *e = getenv("AMPLISOLVE_THREADS")) threads = atoi(e);
        std::vector<std::pair<std::string, std::string>> files;
        {
            PhaseClock::Scope sc("list_files");
            files = list_count_files(a.germline_dir, writer ? list_name : std::string());
        }
        const int total_samples = (int)files.size();
        const std::vector<std::pair<std::string, std::string>> all_files = files; // the whole cohort in visit order (the in-order pass below)
        int first_sample = 0;
        if (sh) files = shard_of_files(files, sh->index, sh->count, &first_sample);
        const int S = (int)files.size();
        std::cout << "\nRunning function storeList: " << list_name << " stored with success. It contains " << total_samples << " samples" << std::endl;
        if (sh) std::cout << "\tshard " << sh->index + 1 << "/" << sh->count << ": samples " << first_sample + 1 << ".." << first_sample + S << std::endl;
        std::cout << "Running function storeGermlineStatistics:" << std::endl;

        // the parsers start NOW, into plain memory, while the runtime is still coming up on the side thread
        std::unique_ptr<ChunkStream> first_stream;
        if (S > 0) first_stream.reset(new ChunkStream(panel, files, threads, false, chunk_bytes_setting(), ring_slots_setting(chunk_bytes_setting())));
        // the host copies of the table are made (their pages touched) while the runtime is still starting, not in front of the download
        const int64_t P = panel.P();
        std::vector<float> rate((size_t)P * 8), germ((size_t)P * 4);
        std::vector<uint8_t> code((size_t)P * 4), gp((size_t)P * 4);
        Dev &dev = dev_async.get();
        float *d_rate = dev.alloc<float>((size_t)P * 8), *d_germ = dev.alloc<float>((size_t)P * 4);
        uint8_t *d_code = dev.alloc<uint8_t>((size_t)P * 4), *d_gp = dev.alloc<uint8_t>((size_t)P * 4);
        int32_t *d_flags = dev.alloc<int32_t>(1);
        dev.check(dev.api->memset_d(dev.ctx, d_flags, 0, sizeof(int32_t)), "memset");
        // The accumulator table the chunks of a streamed cohort are folded into (EE:1057-1481 + the record loop of EE:1484-2544) is
        // streaming state and nothing else here (AMPLI_REDUCE_SUMMARY), so the compact-state kernel may carry it from chunk to chunk;
        // a cohort that arrives as ONE chunk needs no table at all: its launch finalises (one device) or stores slice-major (a shard).
        ampli_acc_table acc{};
        bool have_acc = false;
        auto need_acc = [&] {
            if (have_acc) return;
            void *d_accbuf = dev.alloc<char>(dev.api->acc_bytes(P));
            dev.check(dev.api->acc_bind(d_accbuf, P, &acc), "ampli_acc_bind");
            have_acc = true;
        };
        // a shard's exchange buffers are wanted by its LAST chunk's launch, which writes them (slice-major sums + germ-max pairs)
        void *xbufs[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        if (sh) {
            hook(sh->ee_buffers(sh->user, P, xbufs), "ee_buffers");
            for (void *b : xbufs)
                if (!b) throw Error{AMPLI_E_INVALID, "shard hook ee_buffers returned a null buffer"};
        }
        int launches_compact = 0, launches_compact24 = 0, launches_general = 0;
        void *ev = nullptr;
        dev.check(dev.api->event_create(&ev), "ampli_event_create");
        struct EvGuard { const HipApi *api; void *ev; ~EvGuard() { if (ev) api->event_destroy(ev); } } evg{dev.api, ev};
        DevSlot dslots[kDevSlots];
        int64_t n_lines = 0;
        double parse_s = 0, wait_s = 0, rec_bytes_up = 0;
        int chunks_done = 0;
        // The cohort streams through in chunks of samples: while chunk k is uploaded and reduced into the table, the
        // parser threads are already packing chunks k+1, k+2 into the other pinned buffers.  A depth beyond the fast
        // kernel's integer envelope is only known afterwards (a flag): the cohort then streams a second time through
        // the literal kernel.
        for (int attempt = 0; attempt < 2; ++attempt) {
            chunks_done = 0;
            n_lines = 0;
            rec_bytes_up = 0;
            if (S > 0) {
                std::unique_ptr<ChunkStream> own(attempt == 0 ? first_stream.release() : new ChunkStream(panel, files, threads, false, chunk_bytes_setting(), ring_slots_setting(chunk_bytes_setting())));
                ChunkStream &cs = *own;
                auto next_chunk = [&] {
                    PhaseClock::Scope sc("wait_for_parser");
                    return cs.next();
                };
                for (Chunk *c; (c = next_chunk()) != nullptr;) {
                    if (attempt == 0) // the reference's own message, once per offending line (EE:1178-1181)
                        for (int64_t i = 0; i < c->n_irregular; ++i) std::cout << "malakia paizei edo" << std::endl;
                    const ampli_records r = upload_chunk(dev, dslots[c->slot % kDevSlots], *c, false);
                    const bool fuse = c->last && !sh; // one device holds the whole panel: finalize in the last chunk's launch
                    const bool only = c->last && chunks_done == 0; // the whole cohort (of this shard) in one chunk: no table
                    if (!only) need_acc();
                    const int32_t how = (chunks_done > 0 ? AMPLI_REDUCE_ACCUMULATE : 0) | AMPLI_REDUCE_SUMMARY;
                    {
                        PhaseClock::Scope sc(chunks_done == 0 && attempt == 0 ? "first_launch" : "launch"); // the first one loads the code object
                        if (c->last && sh)
                            dev.check(dev.api->error_reduce_records_sliced(dev.ctx, &r, P, first_sample + c->first, C_value, cov, only ? nullptr : &acc, how,
                                                                           sh->count, (double *)xbufs[0], (float *)xbufs[1]), "ampli_error_reduce_records_sliced");
                        else
                            dev.check(dev.api->error_reduce_records(dev.ctx, &r, P, first_sample + c->first, C_value, cov, only ? nullptr : &acc, how,
                                                                    fuse ? d_rate : nullptr, fuse ? d_code : nullptr, nullptr, fuse ? d_germ : nullptr,
                                                                    fuse ? d_gp : nullptr, fuse ? d_flags : nullptr), "ampli_error_reduce_records");
                        dev.check(dev.api->event_record(dev.ctx, ev), "ampli_event_record");
                        const int which = dev.api->last_reduce_kernel(dev.ctx); // 1 / 2: the compact-state kernel for uint16 / 24-bit records
                        (which == 1 ? launches_compact : which == 2 ? launches_compact24 : launches_general) += 1;
                    }
                    const double w0 = now_s();
                    {
                        PhaseClock::Scope sc("device_wait");
                        dev.check(dev.api->event_sync(ev), "ampli_event_sync"); // the chunk's buffers are free again
                    }
                    wait_s += now_s() - w0;
                    n_lines += c->n_lines;
                    rec_bytes_up += (double)c->n * (double)(P + c->E) * (double)record_bytes(c->layout);
                    ++chunks_done;
                    if ((first_sample + c->first + c->n) / 50 > (first_sample + c->first) / 50)
                        std::cout << "\tParsed successfully " << c->first + c->n << "/" << S << "  samples" << std::endl; // EE:1475-1478
                    cs.release(c);
                }
                parse_s += cs.parse_seconds();
                PhaseClock::add("parser_busy", cs.parse_seconds(), false);
                ChunkStream *done_stream = own.release();
                if (leave_ring_to_exit(a.process_ends)) {
                    done_stream->abandon();
                    delete done_stream;
                } else { // a library caller lives on: unpin + unmap the ring, beside the download and the table writer
                    ring_teardown.run([done_stream] {
                        PhaseClock::Scope sc2("stream_teardown", false);
                        delete done_stream;
                    });
                }
            }
            int32_t kflags = 0;
            {
                PhaseClock::Scope sc("device_wait");
                dev.check(dev.api->ctx_flags(dev.ctx, &kflags, 1), "ampli_ctx_flags");
            }
            if (sh) hook(sh->or_flags(sh->user, &kflags), "or_flags");
            if (!(kflags & AMPLI_FLAG_RERUN_GENERAL) || attempt == 1) break;
            dev.check(dev.api->set_tuning(dev.ctx, 0, 1, 0), "ampli_set_tuning"); // a depth beyond the fast kernel (on some shard): all stream again
            dev.check(dev.api->memset_d(dev.ctx, d_flags, 0, sizeof(int32_t)), "memset");
        }
        double t2 = now_s();
        std::cout << "Running function estimateThresholds: ";
        if (sh) {
            // multi-process run: this shard's table -> position-sliced exchange -> finalize of the own slice -> all-gather
            // -> plane-major table on every shard (include/amplisolve_hip.h, "Position-sliced merge")
            const int n = sh->count;
            const int64_t L = dev.api->slice_len(P, n);
            void **bufs = xbufs;
            if (chunks_done == 0) { // a shard without samples: zero sums, "no qualifying record" germ-max pairs (else: written by the last chunk's launch)
                std::vector<float> none((size_t)n * 8 * L);
                for (int k = 0; k < n; ++k)
                    for (int j = 0; j < 8; ++j)
                        std::fill_n(none.begin() + ((size_t)k * 8 + j) * L, (size_t)L, j < 4 ? -1.0f : -INFINITY);
                dev.check(dev.api->memset_d(dev.ctx, bufs[0], 0, (size_t)n * 21 * L * sizeof(double)), "memset");
                dev.check(dev.api->copy_h2d(dev.ctx, bufs[1], none.data(), none.size() * sizeof(float)), "ampli_copy_h2d");
                dev.sync();
            }
            hook(sh->ee_exchange(sh->user), "ee_exchange");
            dev.check(dev.api->error_finalize_slice(dev.ctx, P, n, sh->index, (const double *)bufs[2], (const float *)bufs[3], C_value, cov,
                                                    bufs[4]), "ampli_error_finalize_slice");
            hook(sh->ee_gather(sh->user), "ee_gather");
            dev.check(dev.api->error_table_unslice(dev.ctx, P, n, bufs[5], d_rate, d_code, nullptr, d_germ, d_gp, d_flags),
                      "ampli_error_table_unslice");
        } else if (chunks_done == 0) {
            throw Error{AMPLI_E_INVALID, "no sample could be read from " + a.germline_dir};
        }
        int32_t flags = 0;
        dev.download(rate.data(), d_rate, rate.size());
        dev.download(code.data(), d_code, code.size());
        dev.download(germ.data(), d_germ, germ.size());
        dev.download(gp.data(), d_gp, gp.size());
        dev.download(&flags, d_flags, 1);
        dev.sync();
        if (flags & 1) {
            // A threshold sum left the exactness envelope (a coverage cut-off of a few reads with depths in the millions): its
            // double is no longer independent of the order of addition, and the reference always writes a table (EE:1597-1606, 1679-1704).
            // So the sums are formed once more in the reference's OWN order -- estimateThresholds' walk of `equal_range`, which libstdc++
            // hands out in reverse insertion order: the last file first, a position's later lines before its first -- one lane per position
            // (ampli_error_sums_inorder).  Every chunk of the WHOLE cohort stays resident for it (the walk starts at the last chunk); the
            // order-free planes (depth sums, counts, Germ_Max) come from an ordinary pass of the literal kernel over the same chunks.  Only
            // the process that writes the table does this (a shard's own table was merged in another order; nobody reads it again).
            std::cout << "\n\ta threshold sum is beyond the range in which its order of addition cannot matter: summing again in the reference's order" << std::endl;
            if (writer) {
                PhaseClock::Scope sc("inorder_pass");
                dev.check(dev.api->set_tuning(dev.ctx, 0, 1, 0), "ampli_set_tuning"); // the literal kernel: every plane exact, any depth
                need_acc();
                ChunkStream cs(panel, all_files, threads, false, chunk_bytes_setting(), ring_slots_setting(chunk_bytes_setting()));
                std::vector<std::unique_ptr<DevSlot>> resident;
                std::vector<ampli_records> descr;
                for (Chunk *c; (c = cs.next()) != nullptr;) {
                    resident.emplace_back(new DevSlot());
                    const ampli_records r = upload_chunk(dev, *resident.back(), *c, false);
                    dev.check(dev.api->error_reduce_records(dev.ctx, &r, P, c->first, C_value, cov, &acc, descr.empty() ? 0 : AMPLI_REDUCE_ACCUMULATE, nullptr, nullptr,
                                                            nullptr, nullptr, nullptr, nullptr), "ampli_error_reduce_records");
                    dev.sync(); // the chunk's host buffer is free again; its device copy stays
                    descr.push_back(r);
                    cs.release(c);
                }
                if (descr.empty()) throw Error{AMPLI_E_INVALID, "no sample could be read from " + a.germline_dir};
                for (size_t k = descr.size(); k-- > 0;)
                    dev.check(dev.api->error_sums_inorder(dev.ctx, &descr[k], P, C_value, cov, &acc, k + 1 == descr.size() ? 0 : 1), "ampli_error_sums_inorder");
                dev.check(dev.api->error_finalize(dev.ctx, &acc, C_value, cov, d_rate, d_code, nullptr, d_germ, d_gp, nullptr), "ampli_error_finalize");
                dev.download(rate.data(), d_rate, rate.size());
                dev.download(code.data(), d_code, code.size());
                dev.download(germ.data(), d_germ, germ.size());
                dev.download(gp.data(), d_gp, gp.size());
                int32_t kf = 0;
                dev.check(dev.api->ctx_flags(dev.ctx, &kf, 1), "ampli_ctx_flags");
                if (kf != 0) throw Error{AMPLI_E_HIP, "the in-order pass raised kernel flags " + std::to_string(kf)};
            }
        }
        double t3 = now_s();

        char name[64];
        snprintf(name, sizeof name, "positionSpecificNoise_%.4f.txt", (double)C_value); // EE:2556
        const std::string out = a.output_dir + "/" + name;
        if (writer) {
            PhaseClock::Scope sc("write_table");
            write_error_table(panel, rate.data(), code.data(), germ.data(), gp.data(), out);
        }
        {
            PhaseClock::Scope sc("join_background");
            interm_files.wait();
            ring_teardown.wait();
        }
        if (sh) hook(sh->barrier(sh->user), "barrier");
        double t4 = now_s();
        std::cout << "\nAmpliSolveErrorEstimation execution was successful. Results can be found at: " << out << std::endl;
        if (getenv("AMPLISOLVE_TIMING"))
            std::cerr << "TIMING panel " << t1 - t0 << "\nTIMING stream " << t2 - t1 << " lines " << n_lines << " chunks " << chunks_done
                      << " parse_busy " << parse_s << " device_wait " << wait_s << " record_MB " << rec_bytes_up / 1e6 << "\nTIMING finish " << t3 - t2 << "\nTIMING write " << t4 - t3
                      // which error_reduce kernel each chunk's launch was (ampli_last_reduce_kernel): error_reduce_u16_kernel / error_reduce_u24_kernel
                      // (compact state) / error_reduce_kernel
                      << "\nTIMING reduce_launches " << launches_compact + launches_compact24 + launches_general << " error_reduce_u16_kernel " << launches_compact
                      << " error_reduce_u24_kernel " << launches_compact24 << " error_reduce_kernel " << launches_general
                      << " accumulator_table " << (have_acc ? 1 : 0) << std::endl;
        std::cout << "\n" << kLine << std::endl;
        return 0;
    } catch (const Error &e) {
        std::cout << "\t\t\nSomething went wrong: " << e.msg << std::endl;
        std::cout << "                                        Sorry but Amplisolve cannot continue..." << std::endl;
        std::cout << kLine << std::endl;
        return e.code ? e.code : -1;
    }
}

// ---------------------------------------------------------------------------------------------------------
namespace {

struct CallRow {
    int sample, line, alt; // sample: index in this process's range of the visit order
    int64_t p;             // panel position
    double q_fw, q_bw;
    float af, af_fw, af_bw;
    int rd, fw, bw, k_fw, k_bw; // the evidence of the call, as the kernel saw it
    int flags;                  // AMPLI_CALL_*
};

} // namespace

int run_variant_calling(const VcArgs &a)
{
    try {
        float p_value = (float)std::atof(a.p_value.c_str()); // VC:262-294
        int cov = std::atoi(a.coverage_cutoff.c_str());
        std::cout << kLine << "\n" << std::endl;
        std::cout << "                          AmpliSolve variant calling for batch execution of multiple samples\n" << std::endl;
        std::cout << "                       MI355X-native build (amplisolve_amd); command line and files as AmpliSolveVariantCalling\n" << std::endl;
        std::cout << "Execution started under the following parameters:" << std::endl;
        std::cout << "\t1. Error estimation                               : " << a.error_file << std::endl;
        std::cout << "\t2. Tumour count dir                               : " << a.tumour_dir << std::endl;
        std::cout << "\t3. Output dir                                     : " << a.output_dir << std::endl;
        if (cov <= 0) {
            cov = 100;
            std::cout << "\t4. Coverage cutoff                                  : User gave: " << a.coverage_cutoff << ". The value is converted to default 100" << std::endl;
        } else {
            std::cout << "\t4. Coverage cutoff                                : " << cov << std::endl;
        }
        if (p_value <= 0 || p_value > 1) {
            p_value = 0.05f;
            std::cout << "\t5. p-value                                         : User gave: " << a.p_value << ". The value is converted to default 0.05" << std::endl;
        } else {
            std::cout << "\t5. p-value                                        : " << p_value << std::endl;
        }
        std::cout << std::endl;

        const ampli_host_shard *sh = (a.shard && a.shard->count > 1) ? a.shard : nullptr;
        NativeShard native;
        if (!sh) {
            native.describe(a.native);
            if (native.active) {
                mkdir_p(a.output_dir); // the default id file lives there
                native.open();
                sh = &native.hooks;
            }
        }
        const bool writer = !sh || sh->index == 0; // shard 0 writes the shared files of a multi-process run
        auto hook = [&](int rc, const char *what) {
            if (rc != 0) throw Error{AMPLI_E_INVALID, std::string("shard hook failed: ") + what + (native.err.empty() ? "" : " -- " + native.err)};
        };
        DevAsync dev_async;
        dev_async.start(); // beside the reading of the error table
        const std::string interm = a.output_dir + "/AmpliSolveVariantCalling_interm_files"; // VC:307
        mkdir_p(writer ? interm : a.output_dir);
        const double t0 = now_s();
        Panel panel;
        std::vector<float> thr;
        Background ring_teardown; // after the panel: joined before it goes away
        {
            PhaseClock::Scope sc("read_table");
            panel_from_error_table(a.error_file, writer ? interm + "/dummyVCF_1.vcf" : std::string(), panel, thr); // VC:320
        }
        std::cout << "Running function storeInputFile: the error levels have stored with success " << panel.walk.size() << std::endl;
        srand((unsigned)time(nullptr));
        const int seed = rand() % 1000;
        const std::string list_name = interm + "/" + std::to_string(seed) + "_tumour_count_list_original.txt"; // VC:332
        int threads = 0;
        if (const char *e = getenv("AMPLISOLVE_THREADS")) threads = atoi(e);
        std::vector<std::pair<std::string, std::string>> files;
        {
            PhaseClock::Scope sc("list_files");
            files = list_count_files(a.tumour_dir, writer ? list_name : std::string());
        }
        const int total_samples = (int)files.size();
        const std::vector<std::pair<std::string, std::string>> all_files = files; // the whole cohort in visit order (the in-order pass below)
        int first_sample = 0;
        if (sh) files = shard_of_files(files, sh->index, sh->count, &first_sample);
        const int T = (int)files.size();
        std::cout << "\nRunning function storeList: " << list_name << " stored with success. It contains " << total_samples << " samples" << std::endl;
        if (sh) std::cout << "\tshard " << sh->index + 1 << "/" << sh->count << ": samples " << first_sample + 1 << ".." << first_sample + T << std::endl;
        std::cout << "\nRunning function callVariants...." << std::endl;

        const double t1 = now_s();
        const int64_t P = panel.P();
        std::vector<CallRow> rows;
        int64_t n_lines = 0;
        double parse_s = 0, rec_bytes_up = 0;
        int chunks_done = 0;
        if (T > 0 || !sh) { // a shard of a multi-process run may hold no tumour file
            // tumour files are independent given the error table: they stream through in chunks (parsing of the next
            // chunks overlaps upload + kernels of this one); only the emitted calls come back.  The parsers start before
            // the context is waited for.
            std::unique_ptr<ChunkStream> own(new ChunkStream(panel, files, threads, true, chunk_bytes_setting(), ring_slots_setting(chunk_bytes_setting())));
            ChunkStream &cs = *own;
            Dev &dev = dev_async.get();
            float *d_thr = dev.upload(thr.data(), thr.size());
            uint8_t *d_ref = dev.upload(panel.ref_code.data(), panel.ref_code.size());
            unsigned long long *d_n = dev.alloc<unsigned long long>(AMPLI_CALL_COUNTER_WORDS);
            DevSlot dslots[kDevSlots];
            auto next_chunk = [&] {
                PhaseClock::Scope sc("wait_for_parser");
                return cs.next();
            };
            for (Chunk *c; (c = next_chunk()) != nullptr;) {
                for (int64_t i = 0; i < c->n_irregular; ++i) std::cout << "malakia paizei edo" << std::endl; // VC:762-765
                const ampli_records r = upload_chunk(dev, dslots[c->slot % kDevSlots], *c, true);
                const int64_t R = P + c->E;
                uint8_t *d_mask = (uint8_t *)dslots[c->slot % kDevSlots].mask.ensure(dev, (size_t)c->n * R + 4);
                int64_t cap = std::max<int64_t>(1 << 16, (int64_t)c->n * R / 16);
                ampli_call *d_calls = nullptr;
                bool done = false;
                std::string why = "call list still overflowing";
                for (int attempt = 0; attempt < 6 && !done; ++attempt) {
                    cap -= cap % AMPLI_CALL_SHARDS;
                    const int64_t per = cap / AMPLI_CALL_SHARDS;
                    if (d_calls) dev.free(d_calls); // the previous attempt's list
                    d_calls = dev.alloc<ampli_call>((size_t)cap);
                    {
                        PhaseClock::Scope sc(chunks_done == 0 && attempt == 0 ? "first_launch" : "launch"); // the first one loads the code object
                        dev.check(dev.api->memset_d(dev.ctx, d_n, 0, sizeof(unsigned long long) * AMPLI_CALL_COUNTER_WORDS), "memset");
                        dev.check(dev.api->poisson_call_records(dev.ctx, &r, P, d_thr, d_ref, cov, AMPLI_POISSON_PREFILTER, d_mask, d_calls, cap, d_n,
                                                                nullptr, nullptr), "ampli_poisson_call_records");
                    }
                    int32_t kflags = 0;
                    {
                        PhaseClock::Scope sc("device_wait");
                        dev.check(dev.api->ctx_flags(dev.ctx, &kflags, 1), "ampli_ctx_flags");
                    }
                    if (kflags & AMPLI_FLAG_QUEUE_OVERFLOW) { // more survivors than the default queue holds: size it for the worst case
                        dev.check(dev.api->set_queue_items(dev.ctx, (int64_t)c->n * R * 3), "ampli_set_queue_items");
                        why = "prefilter queue still overflowing";
                        continue;
                    }
                    std::vector<unsigned long long> n(AMPLI_CALL_COUNTER_WORDS);
                    dev.download(n.data(), d_n, n.size());
                    dev.sync();
                    unsigned long long worst = 0, total = 0;
                    for (int k = 0; k < AMPLI_CALL_SHARDS; ++k) {
                        worst = std::max(worst, n[(size_t)k * AMPLI_CALL_COUNTER_STRIDE]);
                        total += n[(size_t)k * AMPLI_CALL_COUNTER_STRIDE];
                    }
                    if ((int64_t)worst > per) {
                        // a segment overflowed.  Which segment a call lands in depends on the order the workgroups ran in, so
                        // the rerun is sized with headroom: every segment could hold ALL calls of this pass, capped at the
                        // number of (record, alt) pairs there are
                        cap = (int64_t)std::min<unsigned long long>((unsigned long long)c->n * R * 3, std::max(total, 2 * worst)) * AMPLI_CALL_SHARDS;
                        why = "call list still overflowing";
                        continue;
                    }
                    for (int k = 0; k < AMPLI_CALL_SHARDS; ++k) {
                        const size_t cnt = (size_t)n[(size_t)k * AMPLI_CALL_COUNTER_STRIDE];
                        std::vector<ampli_call> calls(cnt);
                        if (cnt) dev.download(calls.data(), d_calls + (size_t)k * per, cnt);
                        dev.sync();
                        for (auto &cl : calls) {
                            const bool prim = cl.record < P;
                            const int line = prim ? c->line_prim[(size_t)cl.sample * P + cl.record] : c->line_ext[(size_t)cl.sample * c->E + (cl.record - P)];
                            const int64_t pp = prim ? (int64_t)cl.record : (int64_t)c->ext_pos[(size_t)(cl.record - P)];
                            rows.push_back(CallRow{c->first + cl.sample, line, cl.alt, pp, cl.q_fw, cl.q_bw, cl.af, cl.af_fw, cl.af_bw, cl.rd, cl.fw,
                                                   cl.bw, cl.k_fw, cl.k_bw, cl.flags});
                        }
                    }
                    done = true;
                }
                if (d_calls) dev.free(d_calls);
                if (!done) throw Error{AMPLI_E_CAPACITY, "variant calling did not complete a pass: " + why};
                n_lines += c->n_lines;
                rec_bytes_up += (double)c->n * (double)(P + c->E) * (double)record_bytes(c->layout);
                ++chunks_done;
                cs.release(c);
            }
            parse_s = cs.parse_seconds();
            PhaseClock::add("parser_busy", parse_s, false);
            ChunkStream *done_stream = own.release();
            if (leave_ring_to_exit(a.process_ends)) {
                done_stream->abandon();
                delete done_stream;
            } else { // a library caller lives on: unpin + unmap the ring, beside the annotation and the writers
                ring_teardown.run([done_stream] {
                    PhaseClock::Scope sc2("stream_teardown", false);
                    delete done_stream;
                });
            }
        }
        const double t2 = now_s();
        // Every emitted pair is scored once more here, with the reference's own operation sequence (score_reference_sequence:
        // kf_gammaq in double with the host's libm, the final log10 in x87 long double, VC:3834-3884), before it is gated,
        // flagged or printed: the device forms Q in fp64 with ROCm's exp / log and agrees to ~1e-10, which decides every pair
        // that is not within 1e-6 of the call gate Q >= 5 (VC:898; those are flagged by the kernel and listed either way) or of
        // the LowQ threshold Q < 20 (VC:1023) -- and since round 5 the PRINTED digits are the host's too, so that no column of
        // the Summary or the VCFs depends on the device's libm.  Sparse (0.1 % of the records), a few threads.
        int64_t n_guarded = 0, n_dropped = 0, n_dropped_unflagged = 0;
        {
            PhaseClock::Scope sc("guard_and_sort");
            std::vector<long double> qf(rows.size()), qb(rows.size());
            {
                int nt_g = (int)std::min<size_t>(std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())), std::max<size_t>(1, rows.size() / 256));
                if (const char *e = getenv("AMPLISOLVE_THREADS")) nt_g = std::max(1, std::min(nt_g, atoi(e)));
                auto score = [&](int tid) {
                    for (size_t i = rows.size() * (size_t)tid / (size_t)nt_g, i1 = rows.size() * (size_t)(tid + 1) / (size_t)nt_g; i < i1; ++i) {
                        const CallRow &c = rows[i];
                        qf[i] = score_reference_sequence(c.k_fw, c.rd - c.bw, thr[(size_t)(0 * 4 + c.alt) * P + c.p]); // VC:895
                        qb[i] = score_reference_sequence(c.k_bw, c.bw, thr[(size_t)(1 * 4 + c.alt) * P + c.p]);        // VC:896
                    }
                };
                std::vector<std::thread> th;
                for (int t = 1; t < nt_g; ++t) th.emplace_back(score, t);
                score(0);
                for (auto &t : th) t.join();
            }
            std::vector<CallRow> kept;
            kept.reserve(rows.size());
            for (size_t i = 0; i < rows.size(); ++i) {
                CallRow &c = rows[i];
                auto near = [](double q, double gate) { return std::fabs(q - gate) <= AMPLI_CALL_GATE_EPS; };
                const bool flagged = (c.flags & AMPLI_CALL_BORDERLINE) || near(c.q_fw, 20) || near(c.q_bw, 20);
                n_guarded += flagged ? 1 : 0;
                if (!(qf[i] >= 5 && qb[i] >= 5)) { // VC:898 in the reference's own arithmetic
                    ++n_dropped;
                    n_dropped_unflagged += flagged ? 0 : 1; // would mean device and host differ by more than the guard's 1e-6: reported below
                    continue;
                }
                c.q_fw = (double)qf[i];
                c.q_bw = (double)qb[i];
                kept.push_back(c);
            }
            if (n_dropped_unflagged)
                std::cerr << "warning: " << n_dropped_unflagged << " pair(s) passed the device's gate by more than 1e-6 and fail the host's; the host's arithmetic decides" << std::endl;
            rows.swap(kept);
            // emission order: samples in visit order, lines in file order, alts in A,C,G,T order (VC:672, 723, 869-3283)
            std::sort(rows.begin(), rows.end(), [](const CallRow &x, const CallRow &y) {
                if (x.sample != y.sample) return x.sample < y.sample;
                if (x.line != y.line) return x.line < y.line;
                return x.alt < y.alt;
            });
        }

        const std::string summary = a.output_dir + "/Summary_Variant_Info.txt"; // VC:342
        // multi-process run: every shard writes its rows to a part file; shard 0 concatenates them in shard order, which is
        // the visit order.  VC:1066 switches the stream to 4 significant digits inside the first row EVER written, so a
        // shard that is not the first to emit starts in that state.
        int64_t before = 0;
        if (sh) hook(sh->rows_before(sh->user, (int64_t)rows.size(), &before), "rows_before");
        std::ofstream output(sh ? summary + ".part" + std::to_string(sh->index) : summary);
        if (before > 0) output << std::setprecision(4);
        if (writer)
        output << "Filename\tChrom\tPosition\tSubtitution\tRD\tRD_fw\tRD_bw\tAF\tReads_fw\tReads_bw\tAF_fw\tAF_bw\tAmpliconEdge_StrandBias\tFisherPvalue\tQscore_fw\tQscore_bw\tReadTier\tGermlineInfo\tMaxGermlineAF\t10merDownstream\t10merUpstream\tHomopolymerFlag" << std::endl; // VC:669
        // Annotation of the emitted calls (Fisher, context, flags: VC:902-1034) and the two text rows of each are independent
        // of every other call: formatted by a few threads, written in order.  VC:1066 sets std::setprecision(4) mid-row and it
        // sticks, so only the first row EVER written (this shard's row 0 when no shard before it emitted) prints its AF columns
        // with the stream's default 6 digits; every VCF row comes from a stream that is still at its default (VC:679).
        std::vector<std::string> sum_line(rows.size()), vcf_line(rows.size());
        {
            PhaseClock::Scope sc("annotate");
            int nt_ann = (int)std::min<size_t>(std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())), std::max<size_t>(1, rows.size() / 64));
            if (const char *e = getenv("AMPLISOLVE_THREADS")) nt_ann = std::max(1, std::min(nt_ann, atoi(e)));
            // AMPLISOLVE_FISHER=off (validation only): leave the statement of VC:902 out, so that p keeps the -1 of VC:901 -- what the
            // reference's own callVariants does when it is compiled without its Fisher statements on a box without Boost
            // (oracle/Makefile, VC_CALL_DROP).  With it every byte of the Summary and the VCF bodies can be compared with that build.
            const char *fisher_env = getenv("AMPLISOLVE_FISHER");
            const bool fisher_off = fisher_env && std::string(fisher_env) == "off";
            auto annotate = [&](int tid) {
                std::ostringstream output, vcf;
                const size_t i0 = rows.size() * (size_t)tid / (size_t)nt_ann, i1 = rows.size() * (size_t)(tid + 1) / (size_t)nt_ann;
                for (size_t i = i0; i < i1; ++i) {
                    const CallRow &c = rows[i];
                    output.str(std::string());
                    vcf.str(std::string());
                    output << std::setprecision((i == 0 && before == 0) ? 6 : 4);
                    const std::string &sample_name = files[(size_t)c.sample].second;
                    const int64_t p = c.p;
                    const std::string &chrom = panel.chroms[panel.pos_chrom[p]];
                    const int pos = panel.pos_coord[p];
                    const int FW = c.fw, BW = c.bw, RD = c.rd; // VC:760-761 and the RD column
                    const int alt_fw = c.k_fw, alt_bw = c.k_bw;
                    const char refc = "ACGT"[panel.ref_code[p]], altc = "ACGT"[c.alt];
                    const std::string Flag_Dup = panel.dup[p] ? "YES" : "NO";
                    const double pf = fisher_off ? -1 : fisher_two_sided(RD - BW, BW, alt_fw, alt_bw); // VC:901-902
                    const std::string Flag_Fisher = pf <= p_value ? "YES" : "NO";         // VC:903-910
                    const std::string Flag_Tier = (alt_fw < 5 || alt_bw < 5) ? "LowQual" : "HighQual"; // VC:912-919
                    const std::string GermlineFlag = "-";                                 // VC:927-935 (map holds a dummy entry only)
                    const std::string MaxGermlineFlag = panel.germ_cell(c.alt, p);        // VC:943-954
                    const std::string down = kmer_down(panel, chrom, pos), up = kmer_up(panel, chrom, pos);
                    const double Q = (c.q_fw + c.q_bw) / 2.000;                           // VC:968
                    const std::string cat = Flag_Dup + "_" + Flag_Fisher;
                    const double max_germ = std::atof(MaxGermlineFlag.c_str());           // VC:972
                    const int homo = homopolymer_test(down, up, altc);
                    // VC:993-1034: flags go through an unordered_map and come out in ITS order
                    std::unordered_map<std::string, std::string> Flag_Hash;
                    int OK = 0;
                    auto put = [&](const char *f) { OK = 1; Flag_Hash.insert(std::make_pair<std::string, std::string>(f, f)); };
                    if (cat == "YES_NO") put("AmpliconEdge");
                    if (cat == "YES_YES") put("AmpliconEdge;StrandBias");
                    if (cat == "NO_YES") put("StrandBias");
                    if (c.af < max_germ && cat == "NO_NO" && Flag_Tier != "HighQual") put("PositionWithHighNoise");
                    if (homo == 1) put("HomoPolymerRegion");
                    if (c.q_fw < 20 || c.q_bw < 20) put("LowQ");
                    if (Flag_Tier != "HighQual") put("LowSupportingReads");
                    std::string filter = "PASS";
                    if (OK) {
                        filter.clear();
                        for (auto it = Flag_Hash.begin(); it != Flag_Hash.end(); ++it) filter += (filter.empty() ? "" : ";") + it->first;
                    }
                    // the C->G block writes "-" instead of "." as ID when the call is not a PASS (VC:1856)
                    const char *id = (!OK || !(refc == 'C' && altc == 'G')) ? "." : "-";
                    vcf << chrom << "\t" << pos << "\t" << id << "\t" << refc << "\t" << altc << "\t" << Q << "\t" << filter << "\t" << c.af << ";" << RD
                        << ";" << alt_fw + alt_bw << "\n"; // VC:1040 / 1062
                    // VC:1066 -- std::setprecision(4) is set mid-row and sticks for every later row of the file
                    output << sample_name << "\t" << chrom << "\t" << pos << "\t" << refc << "->" << altc << "\t" << RD << "\t" << FW << "\t" << BW << "\t"
                           << c.af << "\t" << alt_fw << "\t" << alt_bw << "\t" << c.af_fw << "\t" << c.af_bw << "\t" << Flag_Dup << "_" << Flag_Fisher
                           << "\t" << pf << "\t" << std::setprecision(4) << c.q_fw << "\t" << std::setprecision(4) << c.q_bw << "\t" << Flag_Tier << "\t"
                           << GermlineFlag << "\t" << MaxGermlineFlag << "\t" << down << "\t" << up << "\t" << homo << "\n";
                    sum_line[i] = output.str();
                    vcf_line[i] = vcf.str();
                }
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nt_ann; ++t) th.emplace_back(annotate, t);
            annotate(0);
            for (auto &x : th) x.join();
        }
        PhaseClock::Scope sc_w("write_calls");
        size_t ri = 0;
        for (int t = 0; t < T; ++t) {
            const std::string &sample_name = files[(size_t)t].second;
            std::ofstream vcf(a.output_dir + "/" + sample_name + ".vcf"); // VC:679
            time_t now = time(0);
            char *dt = ctime(&now);
            vcf << "##fileformat=VCF-like\n##fileDate=" << dt
                << "##source=AmpliSolveVariantCalling\n##reference=Not_Specified_here\n##phasing=Not_Specified_here\n##FILTER=<ID=XXXXXXXXX,Description='XXXXXXXXX'>\n##FILTER=<ID=XXXXXXXXX,Description='XXXXXXXXX'>\n##FILTER=<ID=XXXXXXXXX,Description='XXXXXXXXX'>\n##FILTER=<ID=XXXXXXXXX,Description='XXXXXXXXX'>\n##INFO=<ID=RD,Number=1,Type=Integer,Description='Total Read Depth'>\n##SAMPLE=<ID=Not_Specified_here,SampleName="
                << sample_name
                << ">\n##INFO=<ID=AF,Number=.,Type=Float,Description='Allele Frequency'>\n##INFO=<ID=SR,Number=1,Type=String,Description='Supporting Reads'>\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO"
                << std::endl; // VC:688
            if ((t + 1) % 50 == 0) std::cout << "\tParsed successfully " << t + 1 << "/" << T << "  samples" << std::endl;
            for (; ri < rows.size() && rows[ri].sample == t; ++ri) {
                vcf << vcf_line[ri];
                output << sum_line[ri];
            }
        }
        output.close();
        if (sh) {
            hook(sh->barrier(sh->user), "barrier");
            if (writer) {
                std::ofstream all(summary, std::ios::binary);
                for (int k = 0; k < sh->count; ++k) {
                    const std::string part = summary + ".part" + std::to_string(k);
                    std::ifstream in(part, std::ios::binary);
                    if (!in) throw Error{AMPLI_E_INVALID, "missing Summary part of shard " + std::to_string(k) + " (is output_dir shared by all processes?)"};
                    // an empty part (a shard without calls that is not the header's writer) must not touch `all`:
                    // operator<<(streambuf*) sets failbit when it inserts nothing and every later part would be lost
                    if (in.peek() != std::ifstream::traits_type::eof()) all << in.rdbuf();
                    in.close();
                    if (!all.good()) throw Error{AMPLI_E_INVALID, "could not assemble " + summary + " from the shards' parts"};
                    std::remove(part.c_str());
                }
                all.close();
                if (all.fail()) throw Error{AMPLI_E_INVALID, "could not write " + summary};
            }
        }
        ring_teardown.wait();
        if (getenv("AMPLISOLVE_TIMING"))
            std::cerr << "TIMING table " << t1 - t0 << "\nTIMING stream " << t2 - t1 << " lines " << n_lines << " chunks " << chunks_done << " parse_busy "
                      << parse_s << " record_MB " << rec_bytes_up / 1e6 << " calls " << rows.size() << " guarded " << n_guarded << " dropped_by_guard " << n_dropped << "\nTIMING annotate+write " << now_s() - t2 << std::endl;
        std::cout << "\nAmpliSolveVariantCalling execution was successful. The results can be found at : " << summary << std::endl;
        std::cout << "\n" << kLine << std::endl;
        return 0;
    } catch (const Error &e) {
        std::cout << "\t\t\nSomething went wrong: " << e.msg << std::endl;
        std::cout << "                                        Sorry but Amplisolve cannot continue..." << std::endl;
        std::cout << kLine << std::endl;
        return e.code ? e.code : -1;
    }
}

} // namespace ampli
