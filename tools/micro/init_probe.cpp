// tools/micro/init_probe.cpp -- where a fresh process's HIP start-up goes, and what the ways of getting 128 MB of host
// records to the device cost (pinned allocation vs registration vs a plain pageable copy).  Experiment tooling.
//   hipcc -O2 -o tools/micro/init_probe tools/micro/init_probe.cpp ; ./init_probe [hold_MB] [leak]
// hold_MB: also allocate that many MB pinned + device (what a command line holds when it exits); leak: exit without freeing.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define T(name, expr)                                                     \
    do {                                                                  \
        const double t0 = now();                                          \
        hipError_t e = (expr);                                            \
        printf("%-28s %8.2f ms %s\n", name, (now() - t0) * 1e3, e == hipSuccess ? "" : hipGetErrorName(e)); \
    } while (0)

static void touch(char *p, size_t n, int threads)
{
    std::vector<std::thread> th;
    for (int t = 0; t < threads; ++t)
        th.emplace_back([=] {
            for (size_t i = n * t / threads; i < n * (t + 1) / threads; i += 4096) p[i] = (char)i;
        });
    for (auto &x : th) x.join();
}

int main(int argc, char **argv)
{
    const size_t hold = argc > 1 ? (size_t)atoi(argv[1]) << 20 : 0;
    const bool leak = argc > 2;
    const double t_start = now();
    int n = 0;
    T("hipInit", hipInit(0));
    T("hipGetDeviceCount", hipGetDeviceCount(&n));
    T("hipSetDevice", hipSetDevice(0));
    void *d = nullptr, *d2 = nullptr;
    T("hipMalloc 256 B", hipMalloc(&d, 256));
    T("hipMemset", hipMemset(d, 0, 256));
    T("hipMalloc 512 MB", hipMalloc(&d2, 512u << 20));
    const size_t N = 128u << 20;
    char *pg = (char *)aligned_alloc(4096, N);
    double t0 = now();
    touch(pg, N, 16);
    printf("%-28s %8.2f ms\n", "first touch 128 MB (16 thr)", (now() - t0) * 1e3);
    T("H2D pageable 5 MB (first)", hipMemcpy(d2, pg, 5u << 20, hipMemcpyHostToDevice));
    T("H2D pageable 5 MB", hipMemcpy(d2, pg, 5u << 20, hipMemcpyHostToDevice));
    T("H2D pageable 128 MB", hipMemcpy(d2, pg, N, hipMemcpyHostToDevice));
    T("H2D pageable 128 MB again", hipMemcpy(d2, pg, N, hipMemcpyHostToDevice));
    void *pin = nullptr;
    T("hipHostMalloc 128 MB", hipHostMalloc(&pin, N, hipHostMallocDefault));
    t0 = now();
    touch((char *)pin, N, 16);
    printf("%-28s %8.2f ms\n", "touch pinned 128 MB", (now() - t0) * 1e3);
    T("H2D pinned 128 MB", hipMemcpy(d2, pin, N, hipMemcpyHostToDevice));
    T("H2D pinned 128 MB again", hipMemcpy(d2, pin, N, hipMemcpyHostToDevice));
    T("hipHostRegister 128 MB", hipHostRegister(pg, N, hipHostRegisterDefault));
    T("H2D registered 128 MB", hipMemcpy(d2, pg, N, hipMemcpyHostToDevice));
    T("hipHostUnregister", hipHostUnregister(pg));
    T("hipHostFree 128 MB", hipHostFree(pin));
    T("D2H pageable 4 MB", hipMemcpy(pg, d2, 4u << 20, hipMemcpyDeviceToHost));
    void *hp = nullptr, *hd = nullptr;
    if (hold) {
        T("hold: hipHostMalloc", hipHostMalloc(&hp, hold, hipHostMallocDefault));
        T("hold: hipMalloc", hipMalloc(&hd, hold));
        touch((char *)hp, hold, 16);
    }
    printf("%-28s %8.2f ms\n", "total in main", (now() - t_start) * 1e3);
    fflush(stdout);
    if (leak) _exit(0);
    if (hp) T("hold: hipHostFree", hipHostFree(hp));
    if (hd) T("hold: hipFree", hipFree(hd));
    T("hipFree", hipFree(d2));
    fflush(stdout);
    return 0;
}
