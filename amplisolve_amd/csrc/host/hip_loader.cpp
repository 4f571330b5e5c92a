// amplisolve_amd/csrc/host/hip_loader.cpp
#include "hip_loader.hpp"

#include <dlfcn.h>

#include <cstdlib>
#include <mutex>

#include "host.hpp"

namespace ampli {

static HipApi g_api;
static bool g_ok = false;
static std::string g_why;
static std::once_flag g_once;

static std::string own_dir()
{
    Dl_info info;
    if (dladdr((void *)&own_dir, &info) && info.dli_fname) {
        std::string p(info.dli_fname);
        size_t s = p.rfind('/');
        if (s != std::string::npos) return p.substr(0, s);
    }
    return ".";
}

static void load()
{
    std::vector<std::string> cands;
    if (const char *e = getenv("AMPLISOLVE_HIP_LIB")) cands.emplace_back(e);
    const std::string d = own_dir();
    cands.push_back(d + "/libamplisolve_hip.so");
    cands.push_back(d + "/../lib/libamplisolve_hip.so");
    cands.emplace_back("libamplisolve_hip.so");
    void *h = nullptr;
    for (auto &c : cands) {
        h = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (h) break;
        g_why = dlerror();
    }
    if (!h) return;
#define BIND(field, sym)                                                        \
    *(void **)(&g_api.field) = dlsym(h, sym);                                   \
    if (!g_api.field) { g_why = std::string("missing symbol ") + sym; return; }
    BIND(abi_version, "ampli_abi_version") BIND(strerror_, "ampli_strerror") BIND(device_count, "ampli_device_count")
    BIND(ctx_create, "ampli_ctx_create") BIND(ctx_destroy, "ampli_ctx_destroy") BIND(last_error, "ampli_last_error")
    BIND(sync, "ampli_sync") BIND(pinned_alloc, "ampli_pinned_alloc") BIND(pinned_free, "ampli_pinned_free")
    BIND(host_register, "ampli_host_register") BIND(host_unregister, "ampli_host_unregister")
    BIND(dev_alloc, "ampli_dev_alloc") BIND(dev_free, "ampli_dev_free") BIND(copy_h2d, "ampli_copy_h2d")
    BIND(copy_d2h, "ampli_copy_d2h") BIND(memset_d, "ampli_memset_d") BIND(acc_bytes, "ampli_acc_bytes")
    BIND(acc_bind, "ampli_acc_bind") BIND(error_reduce, "ampli_error_reduce") BIND(error_finalize, "ampli_error_finalize") BIND(error_estimate, "ampli_error_estimate")
    BIND(slice_len, "ampli_slice_len") BIND(slice_bytes, "ampli_slice_bytes") BIND(error_reduce_sliced, "ampli_error_reduce_sliced")
    BIND(error_finalize_slice, "ampli_error_finalize_slice") BIND(error_table_unslice, "ampli_error_table_unslice")
    BIND(poisson_call, "ampli_poisson_call") BIND(set_tuning, "ampli_set_tuning") BIND(ctx_flags, "ampli_ctx_flags") BIND(set_queue_items, "ampli_set_queue_items")
    BIND(error_reduce_records, "ampli_error_reduce_records") BIND(poisson_call_records, "ampli_poisson_call_records") BIND(acc_to_slices, "ampli_acc_to_slices")
    BIND(error_reduce_records_sliced, "ampli_error_reduce_records_sliced") BIND(last_reduce_kernel, "ampli_last_reduce_kernel")
    BIND(error_sums_inorder, "ampli_error_sums_inorder")
    BIND(event_create, "ampli_event_create") BIND(event_destroy, "ampli_event_destroy") BIND(event_record, "ampli_event_record") BIND(event_sync, "ampli_event_sync")
    BIND(pileup_count, "ampli_pileup_count")
    BIND(comm_create, "ampli_comm_create") BIND(comm_destroy, "ampli_comm_destroy") BIND(comm_reduce_scatter_f64, "ampli_comm_reduce_scatter_f64")
    BIND(comm_all_to_all_f32, "ampli_comm_all_to_all_f32") BIND(comm_all_gather_bytes, "ampli_comm_all_gather_bytes")
    BIND(comm_all_reduce_max_i32, "ampli_comm_all_reduce_max_i32") BIND(comm_exclusive_sum_i64, "ampli_comm_exclusive_sum_i64") BIND(comm_barrier, "ampli_comm_barrier")
#undef BIND
    if (g_api.abi_version() != AMPLI_ABI_VERSION) { g_why = "libamplisolve_hip.so ABI version mismatch"; return; }
    g_ok = true;
}

const HipApi *hip_api(std::string *why)
{
    std::call_once(g_once, load);
    if (!g_ok && why) *why = g_why;
    return g_ok ? &g_api : nullptr;
}

} // namespace ampli
