// tools/micro/stream_shape.hip -- what the record stream of error_reduce can reach as a memory access SHAPE, with the
// arithmetic replaced by a knob.  Same mapping as error_reduce_kernel: workgroup = 4 waves x 64 positions, wave w streams
// the sample rows [w*S/4, (w+1)*S/4) of its tile (RB bytes per record: a 64*RB-byte segment per row, rows P*RB bytes apart),
// DEPTH rows in flight per lane, WORK dependent-free VALU instructions per row, OCC waves per SIMD (capped through LDS).
//   hipcc --offload-arch=gfx950 -O3 -o stream_shape stream_shape.hip && ./stream_shape
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <vector>

template <int RB> struct Rec;
template <> struct Rec<32> { int4 a, b; };
template <> struct Rec<24> { uint2 a, b, c; };
template <> struct Rec<16> { int4 a; };

template <int RB> __device__ __forceinline__ Rec<RB> ld(const char *q)
{
    Rec<RB> r;
    if constexpr (RB == 32) { r.a = *(const int4 *)q; r.b = *((const int4 *)q + 1); }
    else if constexpr (RB == 24) { const uint2 *u = (const uint2 *)q; r.a = u[0]; r.b = u[1]; r.c = u[2]; }
    else r.a = *(const int4 *)q;
    return r;
}
template <int RB> __device__ __forceinline__ unsigned fold(const Rec<RB> &r)
{
    if constexpr (RB == 32) return r.a.x ^ r.a.y ^ r.a.z ^ r.a.w ^ r.b.x ^ r.b.y ^ r.b.z ^ r.b.w;
    else if constexpr (RB == 24) return r.a.x ^ r.a.y ^ r.b.x ^ r.b.y ^ r.c.x ^ r.c.y;
    else return r.a.x ^ r.a.y ^ r.a.z ^ r.a.w;
}

template <int RB, int DEPTH, int WORK, int OCC>
__global__ __launch_bounds__(256, OCC) void shape_kernel(const char *__restrict__ base, const long long P, const int S, unsigned *out)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    long long p = (long long)blockIdx.x * 64 + lane;
    if (p >= P) p = P - 1;
    const int chunk = S / 4, s0 = wave * chunk;
    const size_t step = (size_t)P * RB;
    const char *q = base + ((size_t)s0 * P + p) * RB;
    Rec<RB> ring[DEPTH];
#pragma unroll
    for (int j = 0; j < DEPTH; ++j) ring[j] = ld<RB>(q + (size_t)(j < chunk ? j : chunk - 1) * step);
    unsigned acc[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    for (int i0 = 0; i0 < chunk; i0 += DEPTH) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            const int i = i0 + j;
            if (i >= chunk) break;
            const unsigned v = fold<RB>(ring[j]);
            const int nx = i + DEPTH < chunk ? i + DEPTH : chunk - 1;
            ring[j] = ld<RB>(q + (size_t)nx * step);
#pragma unroll
            for (int w = 0; w < WORK; ++w) acc[w & 7] = acc[w & 7] * 0x9E3779B1u + v; // v_mad_u32_u24-class work, 8 independent chains
            if (WORK == 0) acc[0] ^= v;
        }
    }
    unsigned r = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) r ^= acc[w];
    if (r == 0x12345678u) out[0] = r; // never (keeps the loads alive)
    if (WORK == 0 && out[15] != 0) atomicXor(&out[1 + (blockIdx.x & 7)], acc[0]); // checksum pass only (out[15] set): 400 k atomics on 8 words cost ~50 us
}

// the same stream through a per-wave LDS ring filled by LDS-DMA (global_load_lds_*), DEPTH rows deep, counted vmcnt
template <int RB, int DEPTH, int WORK, int OCC>
__global__ __launch_bounds__(256) void ring_kernel(const char *__restrict__ base, const long long P, const int S, unsigned *out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLOT = 64 * RB, NI = RB == 16 ? 1 : 2, SZ = RB == 24 ? 12 : 16;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int chunk = S / 4, s0 = wave * chunk;
    const size_t step = (size_t)P * RB;
    const unsigned tile_off = (unsigned)((size_t)blockIdx.x * SLOT), row_end = (unsigned)((size_t)P * RB);
    unsigned v0 = tile_off + lane * SZ, v1 = v0 + 64 * SZ;
    if (v0 + SZ > row_end) v0 = tile_off + (lane * SZ) % RB;
    if (v1 + SZ > row_end) v1 = tile_off + (lane * SZ) % RB;
    const unsigned ring0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(smem + (size_t)wave * DEPTH * SLOT));
    const char *mine = smem + (size_t)wave * DEPTH * SLOT + lane * RB;
    const char *row = base + (size_t)s0 * step;
    auto issue = [&](const char *rp, unsigned dst) {
        unsigned keep;
        if constexpr (RB == 24)
            asm volatile("s_nop 4\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, %3\n\t"
                         "s_add_u32 m0, m0, 768\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %2, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(v0), "v"(v1), "s"(rp), "s"(dst) : "memory");
        else if constexpr (RB == 32)
            asm volatile("s_nop 4\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\t"
                         "s_add_u32 m0, m0, 1024\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(v0), "v"(v1), "s"(rp), "s"(dst) : "memory");
        else
            asm volatile("s_nop 4\n\ts_waitcnt lgkmcnt(0)\n\ts_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(v0), "s"(rp), "s"(dst) : "memory");
    };
    for (int r = 0; r < DEPTH && r < chunk; ++r, row += step) issue(row, ring0 + r * SLOT);
    unsigned acc[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    int slot = 0;
    for (int i = 0; i < chunk; ++i) {
        const int behind = chunk - 1 - i;
        if (behind >= DEPTH - 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NI * (DEPTH - 1)) : "memory");
        else if (DEPTH > 2 && behind == DEPTH - 2) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NI * (DEPTH > 2 ? DEPTH - 2 : 0)) : "memory");
        else if (DEPTH > 3 && behind == DEPTH - 3) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NI * (DEPTH > 3 ? DEPTH - 3 : 0)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const Rec<RB> cur = ld<RB>(mine + slot * SLOT);
        if (i + DEPTH < chunk) { issue(row, ring0 + slot * SLOT); row += step; }
        slot = slot + 1 == DEPTH ? 0 : slot + 1;
        const unsigned v = fold<RB>(cur);
#pragma unroll
        for (int w = 0; w < WORK; ++w) acc[w & 7] = acc[w & 7] * 0x9E3779B1u + v;
        if (WORK == 0) acc[0] ^= v;
    }
    unsigned r = 0;
#pragma unroll
    for (int w = 0; w < 8; ++w) r ^= acc[w];
    if (r == 0x12345678u) out[0] = r;
    if (WORK == 0 && out[15] != 0) atomicXor(&out[1 + (blockIdx.x & 7)], acc[0]);
}

template <int RB, int DEPTH, int WORK, int OCC, bool RING = false> void run(const char *d, long long P, int S, unsigned *d_out, const char *tag)
{
    const unsigned grid = (unsigned)((P + 63) / 64);
    const size_t lds = (size_t)(160 / OCC - 4) * 1024; // dynamic LDS caps the occupancy at OCC workgroups per CU = OCC waves per SIMD
    if (RING && (size_t)4 * DEPTH * 64 * RB > lds) { printf("ring does not fit\n"); return; }
    hipMemset(d_out, 0, 64);
    const unsigned one = 1;
    hipMemcpy(d_out + 15, &one, 4, hipMemcpyHostToDevice); // the first launch also writes the checksum
#define LAUNCH()                                                                                                                  \
    do {                                                                                                                          \
        if (RING) hipLaunchKernelGGL((ring_kernel<RB, DEPTH, WORK, OCC>), dim3(grid), dim3(256), lds, 0, d, P, S, d_out);         \
        else hipLaunchKernelGGL((shape_kernel<RB, DEPTH, WORK, OCC>), dim3(grid), dim3(256), lds, 0, d, P, S, d_out);             \
    } while (0)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    LAUNCH();
    unsigned chk[16];
    hipMemcpy(chk, d_out, 64, hipMemcpyDeviceToHost);
    unsigned cs = 0;
    for (int i = 1; i < 9; ++i) cs = cs * 31 + chk[i];
    hipMemset(d_out, 0, 64);
    for (int i = 0; i < 2; ++i) LAUNCH();
    const int reps = 20;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) LAUNCH();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double bytes = (double)RB * P * S;
    printf("%s %s RB=%d depth=%d work=%3d occ=%d P=%lld: %7.1f us  %5.2f TB/s  checksum %08x\n", tag, RING ? "lds-ring " : "registers", RB, DEPTH, WORK, OCC, P, ms * 1e3,
           bytes / ms / 1e9, WORK == 0 ? cs : 0u);
#undef LAUNCH
    fflush(stdout);
}

int main(int argc, char **argv)
{
    const int S = 256;
    const long long Pmax = 262144;
    char *d = nullptr;
    unsigned *d_out = nullptr;
    const size_t bytes = (size_t)32 * Pmax * S + 4096; // 2.1 GB: far beyond the 256 MiB Infinity Cache
    if (hipMalloc((void **)&d, bytes) != hipSuccess || hipMalloc((void **)&d_out, 64) != hipSuccess) { printf("alloc failed\n"); return 1; }
    { // position-dependent data so that a misplaced LDS image changes the checksum
        std::vector<unsigned> h(1 << 22);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 7);
        for (size_t o = 0; o < bytes; o += h.size() * 4) hipMemcpy(d + o, h.data(), std::min(h.size() * 4, bytes - o), hipMemcpyHostToDevice);
    }
    if (argc > 1 && atoi(argv[1]) == 96) { // poisson_stream's shape: 96 tumour rows, 24 per wave, every wave resident at once
        const long long P = 100000;
        run<24, 1, 0, 4>(d, P, 96, d_out, "S=96 ");
        run<24, 1, 0, 8>(d, P, 96, d_out, "S=96 ");
        run<24, 2, 0, 8>(d, P, 96, d_out, "S=96 ");
        run<24, 4, 0, 8>(d, P, 96, d_out, "S=96 ");
        run<24, 1, 32, 8>(d, P, 96, d_out, "S=96 ");
        run<24, 1, 64, 8>(d, P, 96, d_out, "S=96 ");
        run<24, 4, 32, 8>(d, P, 96, d_out, "S=96 ");
        run<24, 1, 0, 8>(d, P, 256, d_out, "S=256");
        run<24, 1, 32, 8>(d, P, 256, d_out, "S=256");
        // round 4: the same with 16-byte records (what config 3 uploads), seven waves per SIMD like poisson_stream_kernel
        run<16, 1, 0, 7>(d, P, 96, d_out, "S=96 ");
        run<16, 2, 0, 7>(d, P, 96, d_out, "S=96 ");
        run<16, 4, 0, 7>(d, P, 96, d_out, "S=96 ");
        run<16, 1, 32, 7>(d, P, 96, d_out, "S=96 ");
        run<16, 2, 32, 7>(d, P, 96, d_out, "S=96 ");
        run<16, 2, 64, 7>(d, P, 96, d_out, "S=96 ");
        hipFree(d); hipFree(d_out);
        return 0;
    }
    if (argc > 1 && atoi(argv[1]) == 16) { // round 4: the 16-byte stream (1 KB per wave and row) against rows in flight, occupancy and work
        for (const long long P : {4096ll, 81920ll}) { // one workgroup per CU alone; one full round at five waves per SIMD
            run<16, 1, 0, 5>(d, P, S, d_out, "u16  ");
            run<16, 2, 0, 5>(d, P, S, d_out, "u16  ");
            run<16, 3, 0, 5>(d, P, S, d_out, "u16  ");
            run<16, 4, 0, 5>(d, P, S, d_out, "u16  ");
            run<16, 2, 64, 5>(d, P, S, d_out, "u16  ");
            run<16, 3, 64, 5>(d, P, S, d_out, "u16  ");
        }
        for (const long long P : {100000ll, 327680ll}) {
            run<16, 1, 0, 4>(d, P, S, d_out, "u16  ");
            run<16, 2, 0, 4>(d, P, S, d_out, "u16  ");
            run<16, 1, 0, 5>(d, P, S, d_out, "u16  ");
            run<16, 2, 0, 5>(d, P, S, d_out, "u16  ");
            run<16, 3, 0, 5>(d, P, S, d_out, "u16  ");
            run<16, 4, 0, 5>(d, P, S, d_out, "u16  ");
            run<16, 2, 0, 8>(d, P, S, d_out, "u16  ");
            run<16, 4, 0, 8>(d, P, S, d_out, "u16  ");
            run<16, 2, 64, 5>(d, P, S, d_out, "u16  ");
            run<16, 3, 64, 5>(d, P, S, d_out, "u16  ");
            run<16, 2, 96, 5>(d, P, S, d_out, "u16  ");
            run<16, 3, 96, 5>(d, P, S, d_out, "u16  ");
            run<16, 3, 128, 5>(d, P, S, d_out, "u16  ");
        }
        hipFree(d); hipFree(d_out);
        return 0;
    }
    for (int pass = 0; pass < 3; ++pass) {
        if (pass == 1) { hipMemset(d, 1, bytes); printf("-- constant data (every byte 1)\n"); }
        if (pass == 2) { // count-like data: small 24-bit numbers
            std::vector<unsigned> h(1 << 22);
            for (size_t i = 0; i < h.size(); ++i) h[i] = ((unsigned)(i * 2654435761u) >> 22) * 3u;
            for (size_t o = 0; o < bytes; o += h.size() * 4) hipMemcpy(d + o, h.data(), std::min(h.size() * 4, bytes - o), hipMemcpyHostToDevice);
            printf("-- count-like data (10-bit numbers in 32-bit words)\n");
        }
        if (pass == 0) printf("-- random data\n");
        const long long P = 100000;
        run<24, 1, 0, 4>(d, P, S, d_out, "plain");
        run<24, 4, 0, 4, true>(d, P, S, d_out, "plain");
        run<32, 1, 0, 4>(d, P, S, d_out, "plain");
        run<16, 1, 0, 4>(d, P, S, d_out, "plain");
        run<24, 1, 128, 4>(d, P, S, d_out, "work ");
        run<24, 4, 128, 4>(d, P, S, d_out, "work ");
        run<24, 4, 128, 4, true>(d, P, S, d_out, "work ");
        run<24, 1, 0, 4>(d, 131072, S, d_out, "plain");
        run<24, 1, 128, 4>(d, 131072, S, d_out, "work ");
    }
    hipFree(d); hipFree(d_out);
    return 0;
}
