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
void ampli_host_af_limit_batch(const int32_t *d, int64_t n, int32_t *out);
int ampli_host_prefilter_nocall(int32_t k, int32_t rd, float err);
int ampli_host_prefilter_skip_f32(int32_t k, int32_t rd, float err); /* the streaming kernel's fp32 form */

/* ---- panel + cohort: BED (or error table) + a directory of .PILEUP.ASEQ packed into the record SoA ---- */
typedef struct ampli_host_cohort ampli_host_cohort;
const char *ampli_host_last_error(void);
/* bed_or_table: BED panel (EE:615-650) or, with is_error_table != 0, a positionSpecificNoise table (VC:430-576).
 * Reference bases from refbases_file ("chrom pos base" lines, EE:963) or from a FASTA (+ optional .fai);
 * aseq_dir may be NULL (panel only).  Samples come out in the reference's visit order (EE:1081 / VC:672). */
int ampli_host_cohort_load(const char *bed_or_table, int is_error_table, const char *refbases_file, const char *fasta,
                           const char *aseq_dir, int n_threads, int keep_line_no, ampli_host_cohort **out);
void ampli_host_cohort_free(ampli_host_cohort *h);
int64_t ampli_host_cohort_P(const ampli_host_cohort *h);
int64_t ampli_host_cohort_E(const ampli_host_cohort *h);
int32_t ampli_host_cohort_S(const ampli_host_cohort *h);
int64_t ampli_host_cohort_walk_len(const ampli_host_cohort *h);      /* BED-walk rows incl. repeated positions */
const int32_t *ampli_host_cohort_recs(const ampli_host_cohort *h);    /* [S][P+E][8] */
const uint32_t *ampli_host_cohort_dup_off(const ampli_host_cohort *h);/* [P+1] */
const uint32_t *ampli_host_cohort_ext_pos(const ampli_host_cohort *h);/* [E] */
const int32_t *ampli_host_cohort_line_no(const ampli_host_cohort *h); /* [S][P+E] or NULL */
const uint8_t *ampli_host_cohort_ref_code(const ampli_host_cohort *h);/* [P] */
const uint8_t *ampli_host_cohort_dup_flag(const ampli_host_cohort *h);/* [P] */
const char *ampli_host_cohort_sample_name(const ampli_host_cohort *h, int32_t s);
void ampli_host_cohort_stats(const ampli_host_cohort *h, int64_t *lines, int64_t *offpanel, int64_t *irregular, int64_t *malformed);
int ampli_host_position(const ampli_host_cohort *h, int64_t p, char *chrom_out, int chrom_cap, int32_t *coord);
/* sample names of the .ASEQ files of dir in visit order, newline separated; returns the count */
int ampli_host_sample_order(const char *dir, char *out, int64_t cap);

/* ---- the error table on disk (EE:2546-2944 writer, VC:430-576 reader) ---- */
int ampli_host_write_error_table(const ampli_host_cohort *h, const float *rate /*[2][4][P]*/, const uint8_t *code /*[4][P]*/,
                                 const float *germ_val /*[4][P]*/, const uint8_t *germ_present /*[4][P]*/, const char *path);
int ampli_host_read_error_table(const char *path, ampli_host_cohort **out, float *thr_out /*[2][4][P]*/, int64_t thr_capacity);

/* ---- the two command lines as functions (need libamplisolve_hip.so + a GPU) ---- */
int ampli_host_run_error_estimation(const char *panel_design, const char *reference_genome, const char *germline_dir,
                                    const char *C_value, const char *coverage_cutoff, const char *default_error,
                                    const char *output_dir, const char *refbases_file /* NULL: read the FASTA */);
int ampli_host_run_variant_calling(const char *error_file, const char *tumour_dir, const char *output_dir,
                                   const char *coverage_cutoff, const char *p_value);

/* two-sided Fisher exact test of the post-call annotation (VC:3797-3814; own pmf, parity unpinned vs Boost) */
double ampli_host_fisher(int a, int b, int c, int d);

#ifdef __cplusplus
}
#endif
#endif
