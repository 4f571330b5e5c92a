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

/* the same panel as files: BED + "chrom pos base" table, and one .PILEUP.ASEQ per sample (<dir>/<prefix>NNNNN.PILEUP.ASEQ)
 * with exactly the counts ampli_host_synth_fill / ampli_synth_fill produce; returns the bytes of text written (< 0: error) */
int ampli_host_synth_write_panel(const char *bed_path, const char *refbases_path, int64_t P, uint64_t seed);
int64_t ampli_host_synth_write_aseq(const char *dir, const char *prefix, int64_t P, int32_t n_samples, int32_t first_sample,
                                    uint64_t seed, int32_t depth, int32_t tumour, int32_t n_threads);

/* ---- scalar helpers of csrc/ampli_math.h compiled for the host (formatting, unit checks) ---- */
void ampli_host_text_roundtrip_batch(const float *in, int64_t n, float *out);
int32_t ampli_host_af_limit(int32_t d);
void ampli_host_af_limit_batch(const int32_t *d, int64_t n, int32_t *out);
void ampli_host_af_limit_f32_batch(const int32_t *d, int64_t n, int32_t *out); /* ampli_af_limit_f32((float)d[i]) */
int ampli_host_prefilter_nocall(int32_t k, int32_t rd, float err);
int ampli_host_prefilter_skip_f32(int32_t k, int32_t rd, float err); /* the streaming kernel's fp32 form */
/* the all-scores mode's scorer (ampli_poisson_score_dense, csrc/ampli_math.h) on the host, lgamma computed: Q */
void ampli_host_dense_score_batch(const int32_t *k, const int32_t *rd, const float *err, int64_t n, double *q);
/* the drain kernel's scorer (division-free series, csrc/ampli_math.h) for items with k > rd*err > 0: Q and p */
void ampli_host_drain_score_batch(const int32_t *k, const int32_t *rd, const float *err, int64_t n, double *q, double *p);

/* ---- panel + cohort: BED (or error table) + a directory of .PILEUP.ASEQ packed into the record SoA ---- */
typedef struct ampli_host_cohort ampli_host_cohort;
const char *ampli_host_last_error(void);
/* bed_or_table: BED panel (EE:615-650) or, with is_error_table != 0, a positionSpecificNoise table (VC:430-576).
 * Reference bases from refbases_file ("chrom pos base" lines, EE:963) or from a FASTA (+ optional .fai);
 * aseq_dir may be NULL (panel only).  Samples come out in the reference's visit order (EE:1081 / VC:672). */
int ampli_host_cohort_load(const char *bed_or_table, int is_error_table, const char *refbases_file, const char *fasta,
                           const char *aseq_dir, int n_threads, int keep_line_no, ampli_host_cohort **out);
/* the same for shard shard_index of shard_count: only that contiguous range of the visit order is parsed (one process per
 * GPU; the ranges are those of amplisolve_amd/dist.py::shard_range).  first_sample = visit-order index of its sample 0 */
int ampli_host_cohort_load_shard(const char *bed_or_table, int is_error_table, const char *refbases_file, const char *fasta,
                                 const char *aseq_dir, int n_threads, int keep_line_no, int32_t shard_index, int32_t shard_count,
                                 ampli_host_cohort **out);
int32_t ampli_host_cohort_first_sample(const ampli_host_cohort *h);
int32_t ampli_host_cohort_total_samples(const ampli_host_cohort *h);
void ampli_host_cohort_free(ampli_host_cohort *h);
int64_t ampli_host_cohort_P(const ampli_host_cohort *h);
int64_t ampli_host_cohort_E(const ampli_host_cohort *h);
int32_t ampli_host_cohort_S(const ampli_host_cohort *h);
int64_t ampli_host_cohort_walk_len(const ampli_host_cohort *h);      /* BED-walk rows incl. repeated positions */
const int32_t *ampli_host_cohort_recs(const ampli_host_cohort *h);    /* [S][P+E][8] */
const uint32_t *ampli_host_cohort_dup_off(const ampli_host_cohort *h);/* [P+1] */
const uint32_t *ampli_host_cohort_ext_pos(const ampli_host_cohort *h);/* [E] */
const int32_t *ampli_host_cohort_line_no(const ampli_host_cohort *h); /* [S][P+E] or NULL */
/* lines whose RD column is not A+C+G+T: *n entries of four 32-bit words {sample, record slot, occurrence, RD column} */
const uint32_t *ampli_host_cohort_irregular(const ampli_host_cohort *h, int64_t *n);
const uint8_t *ampli_host_cohort_ref_code(const ampli_host_cohort *h);/* [P] */
const uint8_t *ampli_host_cohort_dup_flag(const ampli_host_cohort *h);/* [P] */
const char *ampli_host_cohort_sample_name(const ampli_host_cohort *h, int32_t s);
void ampli_host_cohort_stats(const ampli_host_cohort *h, int64_t *lines, int64_t *offpanel, int64_t *irregular, int64_t *malformed);
int ampli_host_position(const ampli_host_cohort *h, int64_t p, char *chrom_out, int chrom_cap, int32_t *coord);
/* sample names of the .ASEQ files of dir in visit order, newline separated; returns the count */
int ampli_host_sample_order(const char *dir, char *out, int64_t cap);

/* ---- the cohort as the command lines see it: a stream of chunks of consecutive samples, already in the device record
 * layout (what ampli_records of include/amplisolve_hip.h describes).  The callback gets one chunk at a time, in visit
 * order, while the next ones are being parsed; its pointers are valid until it returns.  layout: the narrowest the chunk's
 * counts fit -- AMPLI_RECORDS_U16 (every count <= 65534), AMPLI_RECORDS_U24 (<= 2^24 - 2) or AMPLI_RECORDS_I32; the
 * environment variable AMPLISOLVE_RECORDS = u16 | u24 | i32 sets the narrowest the packer may choose.  prim [n][P] records, ext [n][E] records (E, dup_off,
 * ext_pos are the CHUNK's own); line_prim / line_ext with keep_line_no; irregular: n_irregular x {sample in chunk, record
 * slot, occurrence, RD column} for lines with RD != A+C+G+T.  A negative strand count (reverse above total) or a count
 * beyond int32 ends the stream with AMPLI_E_RANGE.  Return non-zero from the callback to stop. */
typedef int (*ampli_host_chunk_fn)(void *user, int32_t first_sample, int32_t n_samples, int32_t layout, int64_t P, int64_t E,
                                   const void *prim, const void *ext, const uint32_t *dup_off, const uint32_t *ext_pos,
                                   const int32_t *line_prim, const int32_t *line_ext, const uint32_t *irregular, int64_t n_irregular);
int ampli_host_stream_chunks(const ampli_host_cohort *panel, const char *aseq_dir, int n_threads, int keep_line_no,
                             int64_t chunk_bytes, ampli_host_chunk_fn fn, void *user);

/* ---- the error table on disk (EE:2546-2944 writer, VC:430-576 reader) ---- */
int ampli_host_write_error_table(const ampli_host_cohort *h, const float *rate /*[2][4][P]*/, const uint8_t *code /*[4][P]*/,
                                 const float *germ_val /*[4][P]*/, const uint8_t *germ_present /*[4][P]*/, const char *path);
int ampli_host_read_error_table(const char *path, ampli_host_cohort **out, float *thr_out /*[2][4][P]*/, int64_t thr_capacity);
/* the same reader, also writing the by-product file storeInputFile writes (VC:437-441, 571: `chrom \t position \t. ...` per
 * data row, repeated rows included; dummy_vcf NULL or "" = none) */
int ampli_host_read_error_table_vcf(const char *path, const char *dummy_vcf, ampli_host_cohort **out, float *thr_out, int64_t thr_capacity);
/* a table read that way, cell by cell as the caller keeps it -- what storeInputFile puts into its four maps (VC:505-560;
 * the FIRST row of a repeated position wins): which = 0 reference cell (ReferenceBase_Hash), 1..4 threshold cell of A/C/G/T
 * (Thresholds_Hash_Analytic), 5..8 germ-max cell of A/C/G/T (Germline_Max_Hash).  NULL for a panel that did not come from a
 * table or an index out of range; the pointer lives as long as the cohort. */
const char *ampli_host_table_cell(const ampli_host_cohort *h, int64_t p, int32_t which);

/* ---- sequence context of a call (post-call annotation, VC:3307-3718): the 10 reference bases before and after position p
 * of the panel as find_kmer_down / find_kmer_up spell them (a position outside the panel is "-|", or a bare "-" at the
 * offsets -6, -3, -1 and +10) and homopolymerTest's flag for the substituted base `sub`.  down / up: caller's buffers of
 * cap bytes (a cell of a table can be up to 49 characters, so 10 of them need <= 512).  Returns the flag (0 / 1), < 0 on error. */
int ampli_host_context(const ampli_host_cohort *h, int64_t p, char sub, char *down, char *up, int32_t cap);

/* ---- the two command lines as functions (need libamplisolve_hip.so + a GPU) ---- */
int ampli_host_run_error_estimation(const char *panel_design, const char *reference_genome, const char *germline_dir,
                                    const char *C_value, const char *coverage_cutoff, const char *default_error,
                                    const char *output_dir, const char *refbases_file /* NULL: read the FASTA */);
int ampli_host_run_variant_calling(const char *error_file, const char *tumour_dir, const char *output_dir,
                                   const char *coverage_cutoff, const char *p_value);

/* ---- the same command lines as one shard of a multi-process (one process per GPU) run ----
 * Samples shard by contiguous ranges of the visit order (shard k of n takes what amplisolve_amd/dist.py::shard_range
 * gives it); every collective step is a callback, so the transport (torch.distributed over RCCL in
 * amplisolve_amd/multi.py; gloo in the tests) stays outside this library.  Device work is enqueued on the device's
 * default stream and the callbacks must be ordered with it.  Results are byte-identical to the one-process run.
 *   error estimation: position-sliced merge of include/amplisolve_hip.h -- ee_buffers is called once P is known and
 *   returns six DEVICE buffers sized by ampli_slice_bytes(P, count): [0] sums (zeroed), [1] gm, [2] sum_slice,
 *   [3] gm_recv, [4] block (zeroed), [5] blocks; ee_exchange = reduce-scatter SUM [0]->[2] + all-to-all [1]->[3];
 *   ee_gather = all-gather [4]->[5]; or_flags = bitwise OR of a host int over the shards.  Shard 0 writes the files.
 *   variant calling: tumour files are independent; rows_before = exclusive prefix sum of the emitted rows (the
 *   reference's Summary stream changes its float precision after the first row ever written, VC:1066); barrier; shard 0
 *   then assembles Summary_Variant_Info.txt from the shards' parts. */
typedef struct ampli_host_shard {
    int32_t index, count;
    void *user;
    int (*ee_buffers)(void *user, int64_t P, void **d_bufs /*[6]*/);
    int (*ee_exchange)(void *user);
    int (*ee_gather)(void *user);
    int (*or_flags)(void *user, int32_t *flags);
    int (*rows_before)(void *user, int64_t mine, int64_t *before);
    int (*barrier)(void *user);
} ampli_host_shard;
int ampli_host_run_error_estimation_sharded(const char *panel_design, const char *reference_genome, const char *germline_dir,
                                            const char *C_value, const char *coverage_cutoff, const char *default_error,
                                            const char *output_dir, const char *refbases_file, const ampli_host_shard *shard);
int ampli_host_run_variant_calling_sharded(const char *error_file, const char *tumour_dir, const char *output_dir,
                                           const char *coverage_cutoff, const char *p_value, const ampli_host_shard *shard);

/* ---- upstream of the two programs: BAM -> <out_dir>/<bam name without .bam>.PILEUP.ASEQ, i.e. computeCounts / ASEQ PILEUP mode
 * (/root/reference/Execution_examples.md:16-46: vcf= bam= threads= mbq= mrq= mdc= out=; the reference ships it as a binary without
 * source, so parity is unpinned -- DESIGN.md).  The host inflates the BGZF blocks (threads), the device decodes the alignment
 * records and counts (ampli_pileup_count); needs libamplisolve_hip.so + a GPU.  stats (optional) [4]: alignment records, reads
 * kept, bases counted, lines written. */
int ampli_host_compute_counts(const char *vcf, const char *bam, const char *out_dir, int32_t threads, int32_t mbq, int32_t mrq, int32_t mdc,
                              int64_t *stats);
/* the container only (no GPU): stats[4] = alignment records, uncompressed bytes, reference sequences, malformed records (complete
 * records whose CIGAR does not add up to their l_seq or whose fields overrun their block: counted and skipped).  Bytes behind the
 * last complete record -- a truncated or desynchronised stream -- are an error (AMPLI_E_INVALID), not a count: since round 3 the call
 * fails instead of reporting them in stats[3]. */
int ampli_host_bam_scan(const char *bam, int32_t threads, int64_t *stats);

/* the decision guard of the variant-calling command line for calls within 1e-6 of a gate (AMPLI_CALL_BORDERLINE): Q by the
 * reference's own operation sequence (VC:3834-3884: double kf_gammaq, long double log10); *ge5 / *lt20 = the comparisons of
 * VC:898 / VC:1023 made in long double */
double ampli_host_guard_score(int32_t k, int32_t rd, float err, int32_t *ge5, int32_t *lt20);

/* two-sided Fisher exact test of the post-call annotation (VC:3797-3814; own pmf, parity unpinned vs Boost) */
double ampli_host_fisher(int a, int b, int c, int d);
/* the same sum with every term taken from the log-gamma form (slow; the check of the recurrence ampli_host_fisher walks) */
double ampli_host_fisher_direct(int a, int b, int c, int d);

#ifdef __cplusplus
}
#endif
#endif
