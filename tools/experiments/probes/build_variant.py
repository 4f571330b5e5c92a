"""Build experimental variants of libamplisolve_hip.so WITHOUT touching the shipped translation unit: a variant is a list
of (old text, new text) substitutions applied to a copy of csrc/ampli_kernels.hip, compiled into _variants/<name>.so.
Run a tool against one with AMPLISOLVE_HIP_LIB=_variants/<name>.so.   usage: python tools/build_variant.py NAME [NAME ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
SRC = os.path.join(ROOT, "amplisolve_amd", "csrc", "ampli_kernels.hip")
OUT = os.path.join(ROOT, "_variants")

V = {
    # error_reduce without its threshold block / without its Germ_Max block (wrong results; where does the time go?)
    "nothr": [("        if (covok && fw[nt] <= lim_fw && bw[nt] <= lim_bw) { // EE:1595", "        if (covok && fw[nt] == -77 && bw[nt] <= lim_bw) { // variant: never")],
    "nogm": [("        const bool pass = covok && x <= lim_rd; // EE:1251: float(X)/float(RD) <= 0.05", "        const bool pass = covok && x == -77; // variant: never")],
    # poisson_stream at whatever occupancy the register allocator picks (shipped: __launch_bounds__(256, 8))
    "lb256": [("template <int LAY>\n__global__ __launch_bounds__(256, 8) void poisson_stream_kernel(", "template <int LAY>\n__global__ __launch_bounds__(256) void poisson_stream_kernel(")],
    # poisson_stream without clearing the call mask (what do the mask stores cost?)
    "nomask": [("        for (size_t o = w0 + lane; o < w1; o += 64) ((unsigned *)call_mask)[o] = 0u;", "        (void)w1; // variant: mask not cleared")],
    # error_reduce with an XCD-contiguous tile mapping: workgroup b runs on XCD b % 8; give XCD k the k-th contiguous eighth of the
    # tiles (adjacent tiles -> same L2 / same translation) instead of every eighth tile
    "red_xcd": [("    const long long p_raw = (long long)blockIdx.x * W + (lane % W);",
                 "    const unsigned tiles8_ = (gridDim.x + 7u) / 8u;\n    const long long tile_ = (long long)(blockIdx.x & 7u) * tiles8_ + (blockIdx.x >> 3);\n    if (tile_ * W >= P) return;\n    const long long p_raw = tile_ * W + (lane % W);"),
                ("    dim3 grid((unsigned)tiles, (unsigned)splits);", "    dim3 grid((unsigned)((tiles + 7) / 8 * 8), (unsigned)splits);")],
    # TIMING ONLY (wrong results): error_reduce reading a TILE-MAJOR arrangement -- a wave's consecutive sample rows are contiguous
    # (64 records apart) and tiles lie S*64 + PAD records apart -- out of the same buffer.  PAD = 0 / 64 / 192 records.
    "red_tilemajor0": [("    const size_t row_step = (size_t)rv.row_stride * RB; // bytes between the same position of consecutive samples\n    const char *__restrict__ q = rv.base + ((size_t)min(s0, S - 1) * (size_t)rv.row_stride + (size_t)p) * RB;",
                        "    const size_t row_step = (size_t)64 * RB;\n    const char *__restrict__ q = rv.base + ((size_t)blockIdx.x * ((size_t)S * 64 + 0) + (size_t)min(s0, S - 1) * 64 + (size_t)(lane % W)) * RB;")],
    "red_tilemajor64": [("    const size_t row_step = (size_t)rv.row_stride * RB; // bytes between the same position of consecutive samples\n    const char *__restrict__ q = rv.base + ((size_t)min(s0, S - 1) * (size_t)rv.row_stride + (size_t)p) * RB;",
                         "    const size_t row_step = (size_t)64 * RB;\n    const char *__restrict__ q = rv.base + ((size_t)blockIdx.x * ((size_t)S * 64 + 64) + (size_t)min(s0, S - 1) * 64 + (size_t)(lane % W)) * RB;")],
    "red_tilemajor192": [("    const size_t row_step = (size_t)rv.row_stride * RB; // bytes between the same position of consecutive samples\n    const char *__restrict__ q = rv.base + ((size_t)min(s0, S - 1) * (size_t)rv.row_stride + (size_t)p) * RB;",
                          "    const size_t row_step = (size_t)64 * RB;\n    const char *__restrict__ q = rv.base + ((size_t)blockIdx.x * ((size_t)S * 64 + 192) + (size_t)min(s0, S - 1) * 64 + (size_t)(lane % W)) * RB;")],
    # poisson_stream with the prefilter test written branch-free (bitwise & instead of &&: hipcc wraps every short-circuit in an
    # exec-mask block -- s_and_saveexec / s_or exec / s_cbranch_execz around three vector instructions)
    "ps_branchfree": [("            const bool skip_fw = exact && (unsigned)fw[nt] < (unsigned)AMPLI_COUNT_LIMIT && (float)fw[nt] <= c_fw * te[0][nt];\n            const bool skip_bw = exact && (unsigned)bw[nt] < (unsigned)AMPLI_COUNT_LIMIT && (float)bw[nt] <= c_bw * te[1][nt];\n            if (live && nt != ref && !skip_fw && !skip_bw) pushmask |= 1u << nt;",
                       "            const int skip_fw = (int)exact & (int)((unsigned)fw[nt] < (unsigned)AMPLI_COUNT_LIMIT) & (int)((float)fw[nt] <= c_fw * te[0][nt]);\n            const int skip_bw = (int)exact & (int)((unsigned)bw[nt] < (unsigned)AMPLI_COUNT_LIMIT) & (int)((float)bw[nt] <= c_bw * te[1][nt]);\n            pushmask |= (unsigned)((int)live & (int)(nt != ref) & (skip_fw ^ 1) & (skip_bw ^ 1)) << nt;")],
    # poisson_stream at 7 / 6 waves per SIMD (72 / 80 VGPRs): the shipped 8-wave build (64 VGPRs) spills 7-23 registers to scratch,
    # and every wave's spill stores are HBM writes (round 4: that is the "mask touched twice" of the r03 WRITE_SIZE, 17.1 MB vs 9.6 MB)
    # (shipped since round 4: 7; these rebuild the comparison)
    "ps_lb8": [("template <int LAY, bool IRR>\n__global__ __launch_bounds__(256, 7) void poisson_stream_kernel(", "template <int LAY, bool IRR>\n__global__ __launch_bounds__(256, 8) void poisson_stream_kernel(")],
    "ps_lb6": [("template <int LAY, bool IRR>\n__global__ __launch_bounds__(256, 7) void poisson_stream_kernel(", "template <int LAY, bool IRR>\n__global__ __launch_bounds__(256, 6) void poisson_stream_kernel(")],
    # compact uint16 error_reduce kernel (round 4): four waves per SIMD instead of five (what does the occupancy buy on its own?)
    "c_lb4": [("__global__ __launch_bounds__(256, 5) void error_reduce_u16_kernel(", "__global__ __launch_bounds__(256, 4) void error_reduce_u16_kernel(")],
    # (measured on earlier forms of that kernel and folded into it or dropped, DESIGN.md 3.1: the pass condition from the wave mask of
    #  ONE compare -- shipped since; signed cross products with a -1/1 sentinel instead of the "later record" mask -- 114 us, hipcc
    #  keeps two copies of the numerator and branches around the update; two named record sets instead of three -- 107 against 102 us)
    # (three / five sample chunks per workgroup -- 192 / 320 threads, a tree over up to five waves in two LDS slots -- were measured
    #  with a generic form of the kernel's epilogue: 122-125 us / 128-131 us against 108-113 us with four)
    # all-scores kernel (round 5) at three waves per SIMD without spills instead of four with 68-88 bytes of scratch
    "pf_lb3": [("__global__ __launch_bounds__(256, 4) void poisson_full_kernel(", "__global__ __launch_bounds__(256, 3) void poisson_full_kernel(")],
    # (round 5, first form of that kernel -- up to 3 continued-fraction steps in place through the generic loop: 1.03-1.11 ms; every
    #  fraction through the dense list: 1.10-1.17; up to 8 steps in place: 1.16-1.24; three waves per SIMD without spills: 1.16-1.24)
    # ... with the closed form up to k = 16 / k = 8 instead of 4
    "pf_h16": [("#define AMPLI_HORNER_K 4", "#define AMPLI_HORNER_K 16")],
    "pf_h8": [("#define AMPLI_HORNER_K 4", "#define AMPLI_HORNER_K 8")],
    # poisson_stream without queue pushes
    "nopush": [("        if (__any(pushmask != 0)) { // rare", "        if (__any(pushmask != 0) && P < 0) { // variant: never")],
}


def build(name):
    text = open(SRC).read()
    math_h = os.path.join(ROOT, "amplisolve_amd", "csrc", "ampli_math.h")
    math_text = open(math_h).read()
    for old, new in V[name]:
        if old in math_text and old not in text:  # a substitution in the math header: the variant includes its own copy
            alt_h = os.path.join(ROOT, "amplisolve_amd", "csrc", f"_variant_{name}_math.h")
            open(alt_h, "w").write(math_text.replace(old, new))
            text = text.replace('#include "ampli_math.h"', f'#include "_variant_{name}_math.h"')
            continue
        assert old in text, (name, old)
        text = text.replace(old, new)
    os.makedirs(OUT, exist_ok=True)
    src = os.path.join(ROOT, "amplisolve_amd", "csrc", f"_variant_{name}.hip")
    open(src, "w").write(text)
    try:
        others = [os.path.join(ROOT, "amplisolve_amd", "csrc", f) for f in ("ampli_pileup.hip", "ampli_comm.hip")]
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-o",
                        os.path.join(OUT, f"{name}.so"), src, *others], check=True)
    finally:
        os.remove(src)
        alt_h = os.path.join(ROOT, "amplisolve_amd", "csrc", f"_variant_{name}_math.h")
        if os.path.exists(alt_h):
            os.remove(alt_h)


if __name__ == "__main__":
    for n in sys.argv[1:]:
        build(n)
        print("built", n)
