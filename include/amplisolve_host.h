/*
 * include/amplisolve_host.h -- C ABI of libamplisolve_host.so: the C++ host
 * side of the two AmpliSolve command lines (argv, BED / FASTA / .PILEUP.ASEQ /
 * error-table parsing, the SoA packer, the writers).  The executables
 * AmpliSolveErrorEstimation / AmpliSolveVariantCalling are thin mains over
 * this library; tests drive the same entry points through ctypes.
 *
 * The host library holds NO arithmetic of the hot path: sums, rates, p-values
 * and the call gate come from libamplisolve_hip.so (include/amplisolve_hip.h),
 * which it loads at run time and without which every pipeline entry point
 * fails with AMPLI_E_HIP.
 *   EE:n = /root/reference/source_codes/AmpliSolveErrorEstimation.cpp:n
 *   VC:n = /root/reference/source_codes/AmpliSolveVariantCalling.cpp:n
 */
#ifndef AMPLISOLVE_HOST_H
#define AMPLISOLVE_HOST_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic panels: bit-identical to ampli_synth_fill on the device ---- */
int ampli_host_synth_fill(int32_t *recs /*[n_samples][P][8]*/, int64_t P, int32_t n_samples,
                          int32_t first_sample, uint64_t seed, int32_t depth, int32_t tumour);
int ampli_host_synth_ref(uint8_t *ref_code /*[P]*/, int64_t P, uint64_t seed);

/* ---- scalar helpers of csrc/ampli_math.h compiled for the host (formatting, unit checks) ---- */
void ampli_host_text_roundtrip_batch(const float *in, int64_t n, float *out);
int32_t ampli_host_af_limit(int32_t d);
int ampli_host_prefilter_nocall(int32_t k, int32_t rd, float err);

#ifdef __cplusplus
}
#endif
#endif
