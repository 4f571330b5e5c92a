// Issue-rate probe for the instruction classes of the error_reduce inner loop (gfx950): N independent chains per lane,
// one wave per SIMD slot x 4, all CUs.   hipcc --offload-arch=gfx950 -O3 -o op_rates op_rates.hip && ./op_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int iters, unsigned seed)
{
    unsigned a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    const double dd = 1.0 + seed * 1e-9;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (OP == 0) { a0 += a1; a1 += a2; a2 += a3; a3 += a0; }                                  // v_add_u32
            if (OP == 1) { b0 += b1; b1 += b2; b2 += b3; b3 += b0; }                                  // 64-bit integer add
            if (OP == 2) { d0 += dd; d1 += dd; d2 += dd; d3 += dd; }                                  // v_add_f64
            if (OP == 3) { a0 = (a0 & 0xFFFFFF) * (a1 & 0xFFFFFF); a1 = (a1 & 0xFFFFFF) * (a2 & 0xFFFFFF); a2 = (a2 & 0xFFFFFF) * (a3 & 0xFFFFFF); a3 = (a3 & 0xFFFFFF) * (a0 & 0xFFFFFF); } // mul_u32_u24 (+and)
            if (OP == 4) { b0 = (unsigned long long)a0 * 26843545u + b0; b1 = (unsigned long long)a1 * 26843545u + b1; b2 = (unsigned long long)a2 * 26843545u + b2; b3 = (unsigned long long)a3 * 26843545u + b3; a0 += 1; a1 += 1; a2 += 1; a3 += 1; } // v_mad_u64_u32 (+add)
            if (OP == 5) { d0 = fma(d0, dd, d1); d1 = fma(d1, dd, d2); d2 = fma(d2, dd, d3); d3 = fma(d3, dd, d0); } // v_fma_f64
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 + (unsigned long long)(d0 + d1 + d2 + d3);
}

template <int OP> double run(const char *name, int ops_per_iter)
{
    const int blocks = 256 * 4, iters = 4000;
    unsigned long long *d;
    hipMalloc(&d, sizeof(unsigned long long) * blocks * 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<blocks, 256>>>(d, 10, 1);
    hipEventRecord(e0);
    k<OP><<<blocks, 256>>>(d, iters, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: blocks*4 waves / 1024 SIMDs, each iters*8*ops_per_iter instructions
    const double per_simd = (double)blocks * 4 / 1024 * iters * 8 * ops_per_iter;
    const double cyc = ms * 1e-3 * 2.4e9 / per_simd;
    printf("%-28s %7.3f ms  ~%.2f cycles per wave-instruction (at 2.4 GHz; 4.0 = full rate)\n", name, ms, cyc);
    hipFree(d);
    return cyc;
}

int main()
{
    run<0>("v_add_u32", 4);
    run<1>("64-bit integer add", 4);
    run<2>("v_add_f64", 4);
    run<3>("v_mul_u32_u24 (+ v_and)", 8);
    run<4>("v_mad_u64_u32 (+ v_add)", 8);
    run<5>("v_fma_f64", 4);
    return 0;
}
