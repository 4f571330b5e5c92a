/*
 * include/amplisolve_hip.h -- C ABI of libamplisolve_hip.so (MI355X / gfx950).
 *
 * The drop-in boundary of AmpliSolve is its two command lines and their files
 * (SURVEY.md section 8b); the reference exposes no library interface.  This
 * header is the thin C ABI between the C++ host that re-states those command
 * lines (amplisolve_amd/csrc/host) and the hand-written HIP kernels that do the
 * per-position arithmetic.  Every entry point names the reference code it
 * replaces:
 *   EE:n = /root/reference/source_codes/AmpliSolveErrorEstimation.cpp:n
 *   VC:n = /root/reference/source_codes/AmpliSolveVariantCalling.cpp:n
 *
 * Conventions: C linkage, plain pointers and sizes, no exceptions across the
 * boundary.  Every call returns AMPLI_OK (0) or a negative AMPLI_E_* code;
 * ampli_last_error(ctx) gives the detail text.  A context is bound to one
 * device and one HIP stream; all work is enqueued on that stream and is
 * asynchronous unless stated.  The caller owns every buffer.  Pointers named
 * d_* are DEVICE pointers (hipMalloc / ampli_dev_alloc / a torch CUDA tensor).
 * There is no CPU fallback: without a GPU every compute entry point fails
 * with AMPLI_E_HIP.
 *
 * Record arrays (the "count SoA"):
 *   int32 recs[n_samples][R][8],  R = P + E, position index fastest-but-one,
 *   one 32-byte vector {Afw,Cfw,Gfw,Tfw,Ars,Crs,Grs,Trs} per (sample, record):
 *   a dense (position x strand x base) tensor per sample.  Xfw = X - Xrs
 *   (EE:1155-1158, VC:767-770), Xrs = the ASEQ reverse-strand column.
 *   Absent record (the sample's file has no line for the position):
 *   recs[..][0] == AMPLI_ABSENT.
 *   P = unique panel positions.  Record r < P is the first line of position r
 *   in the sample's file.  E = extra occurrences: a position covered by
 *   overlapping amplicons is listed again in every ASEQ file and each line is
 *   a record of its own (it counts in the quorum, the sums and the germ-max
 *   sequence).  Extras of position p are records P+dup_off[p] ..
 *   P+dup_off[p+1]-1 and are visited right after record (s,p) (file order);
 *   ext_pos[e] is the position of extra e.  E = 0: pass NULL for both.
 *   Samples are in the reference's visit order (EE:1081 / VC:672).
 *   Counts must be < 2^24 (AMPLI_E_RANGE is reported by the host packer).
 */
#ifndef AMPLISOLVE_HIP_H
#define AMPLISOLVE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMPLI_ABI_VERSION 5
#define AMPLI_ABSENT INT32_MIN
/* record layouts, same field order in all of them:
 *   AMPLI_RECORDS_I32  int32 recs[n_samples][R][8], 32 B per record, absent: recs[..][0] == INT32_MIN
 *   AMPLI_RECORDS_U24  8 x 24-bit little-endian fields, 24 B per record, absent: field 0 == 0xFFFFFF; for counts
 *                      <= 2^24 - 2, which is every count the fast kernels accept anyway (a quarter fewer HBM bytes)
 *   AMPLI_RECORDS_U16  uint16 recs[n_samples][R][8], 16 B per record, absent: recs[..][0] == 0xFFFF; for cohorts
 *                      whose every count is <= 65534 (half the HBM bytes) */
#define AMPLI_RECORDS_I32 0
#define AMPLI_RECORDS_U16 1
#define AMPLI_RECORDS_U24 2

#define AMPLI_OK 0
#define AMPLI_E_INVALID (-1)  /* bad argument */
#define AMPLI_E_HIP (-2)      /* HIP runtime error / no device */
#define AMPLI_E_NOMEM (-3)
#define AMPLI_E_ENVELOPE (-4) /* double accumulators left the exactness envelope */
#define AMPLI_E_CAPACITY (-5) /* compact call list overflowed; n_calls holds the needed size */
#define AMPLI_E_RANGE (-6)    /* a count does not fit the kernels' integer envelope */
#define AMPLI_E_COMM_TIMEOUT (-7) /* ampli_comm_create: ncclCommInitRank did not return in time.  A helper thread is still inside
                                   * RCCL on this device: the process MUST end now -- print, flush, _exit(1); do not destroy the
                                   * context, do not run static destructors, never re-exec */

typedef struct ampli_ctx ampli_ctx;

int ampli_abi_version(void);
const char *ampli_strerror(int code);
/* number of HIP devices visible; 0 when there is none (never initialises a device) */
int ampli_device_count(void);
/* the same with the reason when the answer is not a count: the number of devices, or -1 with hipGetDeviceCount's error
 * name and text in msg (cap bytes) */
int ampli_device_probe(char *msg, size_t cap);

/* stream: the hipStream_t to enqueue on (e.g. torch's current stream); NULL is the
 * device's default (null) stream; AMPLI_STREAM_OWN lets the context create and own a
 * non-blocking stream (which does not synchronise with the null stream). */
#define AMPLI_STREAM_OWN ((void *)(intptr_t)-1)
int ampli_ctx_create(int device_ordinal, void *stream, ampli_ctx **out);
/* layout of every record array (d_recs / d_trecs) handed to this context from now on; results are identical */
int ampli_set_record_layout(ampli_ctx *ctx, int32_t layout);
/* device-side conversion 8 x int32 -> 8 x uint16 of n_records records; *d_overflow is OR-ed with 1 when a count does
 * not fit (the caller then stays with AMPLI_RECORDS_I32) */
int ampli_records_pack16(ampli_ctx *ctx, const int32_t *d_recs32, int64_t n_records, void *d_recs16, int32_t *d_overflow);
/* the same for the 24-byte layout (8 x 24 bits); overflow: a count above 2^24 - 2 */
int ampli_records_pack24(ampli_ctx *ctx, const int32_t *d_recs32, int64_t n_records, void *d_recs24, int32_t *d_overflow);
void ampli_ctx_destroy(ampli_ctx *ctx);
const char *ampli_last_error(ampli_ctx *ctx);
int ampli_sync(ampli_ctx *ctx);        /* hipStreamSynchronize */
void *ampli_stream(ampli_ctx *ctx);    /* the hipStream_t in use */

/* memory plumbing */
int ampli_pinned_alloc(size_t bytes, void **out);
int ampli_pinned_free(void *p);
/* pin (page-lock + map for the device) host memory the caller owns, e.g. a record buffer the parsers filled while the runtime
 * was still starting; p / bytes page aligned; ctx (may be NULL) names the device.  ampli_host_unregister before the memory is freed. */
int ampli_host_register(ampli_ctx *ctx, void *p, size_t bytes);
int ampli_host_unregister(void *p);
int ampli_dev_alloc(ampli_ctx *ctx, size_t bytes, void **d_out);
int ampli_dev_free(ampli_ctx *ctx, void *d_p);
int ampli_copy_h2d(ampli_ctx *ctx, void *d_dst, const void *src, size_t bytes); /* async */
int ampli_copy_d2h(ampli_ctx *ctx, void *dst, const void *d_src, size_t bytes); /* async */
int ampli_memset_d(ampli_ctx *ctx, void *d_dst, int byte, size_t bytes);        /* async */

/* events on the context's stream (HIP events; bench.py times kernels with these) */
int ampli_event_create(void **ev);
int ampli_event_destroy(void *ev);
int ampli_event_record(ampli_ctx *ctx, void *ev);
int ampli_event_sync(void *ev);                                         /* hipEventSynchronize */
int ampli_event_elapsed_ms(void *ev_start, void *ev_stop, float *ms); /* synchronises on ev_stop */

/*
 * Accumulator table: what storeGermlineStatistics + the record loop of
 * estimateThresholds leave behind per (position, nucleotide), as planes with the
 * position index fastest.  All planes live in one buffer of ampli_acc_bytes(P)
 * bytes; ampli_acc_bind carves the pointers (buffer order: snt, srd, cnt, nrec, gm_n, gm_first_af, gm_rest,
 * gm_first).  The planes:
 *   snt  double [2][4][P]  sum of X_s + float(RD_s)*float(C) over qualifying records   (EE:1597,1599)
 *   srd  int64  [2][4][P]  sum of RD_s over qualifying records                         (EE:1598,1600)
 *   cnt  int32  [4][P]     qualifying records                                          (EE:1606)
 *   nrec int32  [P]        all records of the position = Value_Hash.count(key)         (EE:1659)
 *   gm_n int32  [4][P]     records qualifying for Germ_Max                             (EE:1251)
 *   gm_first    int32 [4][P]  global sample index of the first qualifying record (INT32_MAX: none; -1: unknown,
 *                             after a gathered merge -- bookkeeping only, last plane of the buffer)
 *   gm_first_af float [4][P]  its AF (the reference discards it, EE:1258-1261; kept to merge shards)
 *   gm_rest     float [4][P]  max AF over the later qualifying records (-inf: none)    (EE:1263-1270)
 * [2] = strand (0 forward, 1 reverse), [4] = nucleotide A,C,G,T.
 * snt|srd|cnt|nrec|gm_n are plain sums over samples: shards merge by addition
 * (RCCL all-reduce SUM per dtype); the gm_* triple merges in sample order
 * (ampli_acc_merge).
 */
typedef struct ampli_acc_table {
    int64_t P;
    double *snt;
    int64_t *srd;
    int32_t *cnt;
    int32_t *nrec;
    int32_t *gm_n;
    int32_t *gm_first;
    float *gm_first_af;
    float *gm_rest;
} ampli_acc_table;

size_t ampli_acc_bytes(int64_t P);
int ampli_acc_bind(void *base, int64_t P, ampli_acc_table *out);

/*
 * error_reduce -- replaces the per-line derivation of storeGermlineStatistics
 * (EE:1149-1232), its running Germ_Max (EE:1251-1296 + C/G/T clones) and the
 * gate/accumulate loop of estimateThresholds (EE:1565-1631 + clones) for S
 * samples.  first_sample = global index of sample 0 of this shard.
 * C, coverage_cutoff as the reference's C_value / coverage_cutoff after its
 * defaulting (EE:372-388).  d_acc: device table (fully overwritten).
 */
int ampli_error_reduce(ampli_ctx *ctx, const int32_t *d_recs, int64_t P, int64_t E,
                       const uint32_t *d_dup_off, int32_t S, int32_t first_sample, float C,
                       int32_t coverage_cutoff, const ampli_acc_table *d_acc);

/*
 * error_estimate -- error_reduce + error_finalize in one call for a panel that lives on ONE device (the
 * single-GPU command line): the merged per-position state is finalised in the reduce kernel's epilogue, so the
 * accumulator table never travels through HBM (d_acc may be NULL; pass a table to get it as well).  Outputs as
 * for ampli_error_finalize.  Falls back to reduce -> merge -> finalize internally when the sample axis has to be
 * split across workgroups (small panels).
 */
int ampli_error_estimate(ampli_ctx *ctx, const int32_t *d_recs, int64_t P, int64_t E, const uint32_t *d_dup_off,
                         int32_t S, float C, int32_t coverage_cutoff, const ampli_acc_table *d_acc, float *d_rate,
                         uint8_t *d_code, float *d_thr, float *d_germ_val, uint8_t *d_germ_present, int32_t *d_flags);

/*
 * A cohort on the device described explicitly -- what the streaming command lines use: a cohort is uploaded in
 * CHUNKS of consecutive samples (visit order) while the next chunk is still being parsed, each chunk in its own
 * buffers and, if need be, its own layout (24-byte records unless a count of the chunk needs int32).
 *   recs        DEVICE, primary records [n_samples][row_stride] in `layout`
 *   row_stride  records between the same position of consecutive samples; 0 = dense (P when `ext` is given, else
 *               P + E).  A padded stride keeps panels whose row is a large power of two off the same HBM channels.
 *   ext         DEVICE, extra-occurrence records [n_samples][ext_stride]; NULL = they follow the primaries inside
 *               each row (the dense interchange layout: recs + P records)
 *   ext_stride  0 = E
 *   E, dup_off [P+1] (error_reduce), ext_pos [E] (poisson_call): as in the classic entry points; a chunk may carry
 *               its own E / dup_off / ext_pos (slots for the multiplicities seen in ITS files)
 *   rd, rd_ext  optional RD column of irregular lines (below)
 */
typedef struct ampli_records {
    const void *recs;
    int64_t row_stride;
    const void *ext;
    int64_t ext_stride;
    int64_t E;
    const uint32_t *dup_off;
    const uint32_t *ext_pos;
    int32_t layout;
    int32_t n_samples;
    /* optional: the RD column of the lines whose RD differs from A+C+G+T (EE:1178-1181, VC:762-765).  The reference
     * goes on to use the column: as the denominator of the Germ_Max AF (EE:1229-1232) and, in the caller, in AF = X/RD
     * and in the forward depth RD - RD_reverse of the Poisson test (VC:814-817, VC:895).  DEVICE int32 rd [n_samples][P],
     * rd_ext [n_samples][E], AMPLI_ABSENT where the line is regular; NULL = every line of the cohort is regular (the
     * fast kernels; a cohort with an RD plane goes through the literal reduce kernel). */
    const int32_t *rd;
    const int32_t *rd_ext;
} ampli_records;

/*
 * error_reduce over one chunk (EE:1149-1296, EE:1565-1631 as ampli_error_reduce).  accumulate == 0: d_acc is
 * overwritten (first chunk); accumulate != 0: d_acc holds the state of the EARLIER samples and receives
 * d_acc (+) chunk -- sums add, the Germ_Max triple composes in sample order, so any chunking of the visit order
 * gives the table of the single pass.  With d_rate / d_code non-NULL the merged state is finalised in the same
 * launch (last chunk; outputs as ampli_error_finalize).  Fully asynchronous on the context's stream.
 */
/* `accumulate` is a set of bits (0 / 1 as before):
 *   AMPLI_REDUCE_ACCUMULATE  d_acc holds the state of the earlier samples (above)
 *   AMPLI_REDUCE_SUMMARY     the caller takes d_acc as STREAMING STATE, not as the bookkeeping of record: gm_n may then count a
 *                            chunk's qualifying records as none / one / two-or-more and gm_first may read -1 (unknown) -- what every
 *                            merge, the sliced store and finalize ask of them is kept, every other plane is exact.  That is what lets
 *                            the compact-state kernel (ampli_set_reduce_compact) carry a streamed uint16 cohort from chunk to chunk;
 *                            chunks that take the general kernel (other layouts, lines with their own RD column) leave exact planes
 *                            and the two compose in either order.  The command lines stream with this bit. */
#define AMPLI_REDUCE_ACCUMULATE 1
#define AMPLI_REDUCE_SUMMARY 2
int ampli_error_reduce_records(ampli_ctx *ctx, const ampli_records *recs, int64_t P, int32_t first_sample, float C,
                               int32_t coverage_cutoff, const ampli_acc_table *d_acc, int32_t accumulate, float *d_rate,
                               uint8_t *d_code, float *d_thr, float *d_germ_val, uint8_t *d_germ_present, int32_t *d_flags);
/* The last chunk of a SHARD's streamed cohort (one process per GPU): d_acc (+) chunk goes straight into the slice-major
 * exchange buffers of the position-sliced merge below (what ampli_error_reduce_sliced writes for a resident shard), and into
 * d_acc as well when one is given.  d_acc may be NULL when the shard's cohort is this one chunk (accumulate == 0). */
int ampli_error_reduce_records_sliced(ampli_ctx *ctx, const ampli_records *recs, int64_t P, int32_t first_sample, float C,
                                      int32_t coverage_cutoff, const ampli_acc_table *d_acc, int32_t accumulate, int32_t n_slices,
                                      double *d_sums, float *d_gm);

/*
 * The eight threshold sums of d_acc (snt[2][4][P]) in the REFERENCE's own order of addition -- what estimateThresholds'
 * walk of `equal_range` (EE:1555-1606) amounts to with libstdc++: the last file of the visit order first, a position's later
 * lines of a file before its first one -- for cohorts outside the exactness envelope (ampli_error_finalize raised flag bit 0:
 * a sum's partial sums are no longer all exact, so the double depends on the order).  recs = one chunk; a cohort in several
 * chunks is walked from its LAST chunk (accumulate = 0) to its first (accumulate = 1).  Every other plane of d_acc is left as
 * an ordinary ampli_error_reduce_records pass wrote it (they do not depend on the order); ampli_error_finalize on the table then
 * gives the reference's rates (it raises flag bit 0 again: pass d_flags = NULL or ignore the bit).  One lane per position,
 * sequential over the samples: a fallback, not a fast path.
 */
int ampli_error_sums_inorder(ampli_ctx *ctx, const ampli_records *recs, int64_t P, float C, int32_t coverage_cutoff,
                             const ampli_acc_table *d_acc, int32_t accumulate);

/*
 * acc_merge -- ordered combine of nparts partial tables (parts[0] = earliest
 * samples) into d_dst (may alias parts[0]).  Sums add; the germ-max triple
 * composes as the reference's sequential state machine would.  This is the
 * local half of the multi-GPU merge and of in-GPU sample splitting.
 */
int ampli_acc_merge(ampli_ctx *ctx, const ampli_acc_table *d_dst, const ampli_acc_table *d_parts,
                    int32_t nparts);

/* Byte regions of a table buffer for the multi-GPU merge: [0, sum_bytes) holds the planes that merge by
 * addition (snt | srd | cnt | nrec | gm_n; reduce each with its own dtype), [gm_offset, gm_offset+gm_bytes)
 * holds gm_n | gm_first_af | gm_rest, 48*P bytes (all-gather it BEFORE gm_n is reduced). */
int ampli_acc_regions(int64_t P, size_t *sum_bytes, size_t *gm_offset, size_t *gm_bytes);

/* The additive planes as ONE float64 buffer [snt 8P | srd 8P | cnt 4P | nrec P] (ampli_acc_packed_len(P) = 21*P
 * doubles) so that the shards merge with a single all-reduce (SUM); the integers are exact in a double. */
int64_t ampli_acc_packed_len(int64_t P);
int ampli_acc_pack(ampli_ctx *ctx, const ampli_acc_table *d_acc, double *d_packed);
int ampli_acc_unpack(ampli_ctx *ctx, const double *d_packed, const ampli_acc_table *d_acc);

/* Multi-GPU fast path (same results as reduce -> pack ... unpack -> gm_merge -> finalize, two launches fewer and no
 * table round trip): error_reduce_packed writes a shard's additive planes straight into the all-reduce buffer
 * (d_acc then only receives the germ-max planes, for the all-gather); error_finalize_merged finalises from the
 * all-reduced buffer and the gathered germ-max regions (rank order = sample order). */
int ampli_error_reduce_packed(ampli_ctx *ctx, const int32_t *d_recs, int64_t P, int64_t E, const uint32_t *d_dup_off,
                              int32_t S, int32_t first_sample, float C, int32_t coverage_cutoff,
                              const ampli_acc_table *d_acc, double *d_packed);
int ampli_error_finalize_merged(ampli_ctx *ctx, int64_t P, const double *d_packed, const void *d_gm_regions,
                                int32_t nparts, float C, int32_t coverage_cutoff, float *d_rate, uint8_t *d_code,
                                float *d_thr, float *d_germ_val, uint8_t *d_germ_present, int32_t *d_flags);

/* Position-sliced merge (the default multi-GPU exchange; ~2.5x less xGMI traffic than all-reduce + all-gather of whole
 * tables).  With n ranks, rank k owns the positions [k*L, (k+1)*L), L = ampli_slice_len(P, n) (ceil(P/n) rounded up to
 * 64).  Per batch and rank:
 *   ampli_error_reduce_sliced   shard of the normal panel -> d_sums f64 [n][21][L] (planes snt 8 | srd 8 | cnt 4 | nrec 1,
 *                               exact integers in doubles) and d_gm f32 [n][8][L] (germ-max of the shard: first AF of
 *                               EE:1251-1261, -1 when no record qualified | max of the later ones, -inf when none);
 *                               entries of positions >= P are never written (allocate the buffers zeroed)
 *   reduce-scatter(SUM) of d_sums -> the rank's [21][L]; all-to-all of d_gm -> [n][8][L] (chunk k = rank k's pair)
 *   ampli_error_finalize_slice  quorum / rate / NaN / text round trip / Germ_Max sentinel (EE:1659-1714, EE:1260...) of
 *                               the slice, the germ-max pairs folded in rank order = sample order -> one block
 *   all-gather of the blocks    (block = rate f32[8][L] | thr f32[8][L] | germ_val f32[4][L] | code u8[4][L] |
 *                               germ_present u8[4][L] | 64-byte tail holding the int32 exactness flags of the slice; the flag word
 *                               is OR-ed into, like every flag of this library: allocate the block zeroed)
 *   ampli_error_table_unslice   blocks -> the plane-major error table ([2][4][P] ...) every other entry point uses;
 *                               ORs the blocks' flags into *d_flags
 * Results are bit-identical to ampli_error_estimate over all shards in order. */
/* Several independent batches can share ONE round of collectives (fewer, larger messages; fewer cross-stream waits):
 * with ampli_set_slice_group(ctx, G, g) the exchange buffers hold G batches per slice chunk -- d_sums
 * [n][G][21][L], d_gm [n][G][8][L], the reduce-scattered / all-to-all'ed / gathered buffers accordingly [G][21][L],
 * [n][G][8][L], [G][block], [n][G][block] -- and every call of the four entry points below (and of
 * ampli_poisson_call_blocks) addresses batch g of the group.  Default G = 1, g = 0. */
int ampli_set_slice_group(ampli_ctx *ctx, int32_t group_size, int32_t group_index);
int64_t ampli_slice_len(int64_t P, int32_t n_slices);
int ampli_slice_bytes(int64_t P, int32_t n_slices, size_t *sums_bytes, size_t *gm_bytes, size_t *block_bytes);
/* The sums' format (round 4).  AMPLI_SLICE_WIDE (default): the 21 planes above, 168 B per position.  AMPLI_SLICE_SLIM: 14 planes,
 * 112 B per position, the same ONE reduce-scatter of doubles: snt 8 | srd 4 (the two strands' depth sums of a nucleotide in one
 * double: forward + reverse * 2^26) | cnt0 + cnt1 * 2^17 + cnt2 * 2^34 | cnt3 + nrec * 2^17.  Adding doubles adds the fields
 * independently and exactly while every field's total stays below its width, so each of the n shards may use 1/n of a field's
 * range (depth sums < 2^26 / n, counts < 2^17 / n per position): ampli_error_reduce_sliced / ampli_acc_to_slices check that where
 * they pack and raise AMPLI_FLAG_SLICE_RANGE otherwise -- repeat the exchange in the wide format then.  The format applies to
 * every sliced entry point of the context (the [21] in the shapes above becomes ampli_slice_planes(format)); results are
 * bit-identical in both. */
#define AMPLI_SLICE_WIDE 0
#define AMPLI_SLICE_SLIM 1
int ampli_set_slice_format(ampli_ctx *ctx, int32_t format);
int32_t ampli_slice_planes(int32_t format);
int ampli_slice_bytes_fmt(int64_t P, int32_t n_slices, int32_t format, size_t *sums_bytes, size_t *gm_bytes, size_t *block_bytes);
int ampli_error_reduce_sliced(ampli_ctx *ctx, const int32_t *d_recs, int64_t P, int64_t E, const uint32_t *d_dup_off,
                              int32_t S, int32_t first_sample, float C, int32_t coverage_cutoff, int32_t n_slices,
                              double *d_sums, float *d_gm);
/* a shard's accumulator table (e.g. built chunk by chunk with ampli_error_reduce_records) -> the slice-major exchange
 * buffers ampli_error_reduce_sliced would have written */
int ampli_acc_to_slices(ampli_ctx *ctx, const ampli_acc_table *d_acc, int32_t n_slices, double *d_sums, float *d_gm);
int ampli_error_finalize_slice(ampli_ctx *ctx, int64_t P, int32_t n_slices, int32_t slice_index,
                               const double *d_sum_slice, const float *d_gm_recv, float C, int32_t coverage_cutoff,
                               void *d_block);
int ampli_error_table_unslice(ampli_ctx *ctx, int64_t P, int32_t n_slices, const void *d_blocks, float *d_rate,
                              uint8_t *d_code, float *d_thr, float *d_germ_val, uint8_t *d_germ_present, int32_t *d_flags);

/* gm_merge -- fold nparts gathered gm regions (region k = shard k, ascending sample order, laid out
 * back to back, gm_bytes each) into d_dst's germ-max planes.  The sequential state machine of
 * EE:1251-1271 composes over shards exactly as ampli_acc_merge does. */
int ampli_gm_merge(ampli_ctx *ctx, const ampli_acc_table *d_dst, const void *d_regions, int32_t nparts);

/*
 * error_finalize -- replaces the quorum/divide/NaN logic of estimateThresholds
 * (EE:1659-1714 + clones) and the Germ_Max sentinel rule (EE:1260,1318,1374,1431).
 *   d_rate [2][4][P] float : float(snt)/float(srd)                       (EE:1679-1680)
 *   d_code [4][P]    uint8 : 0 estimate, 1 below quorum (EE:1659), 2 NaN (EE:1682)
 *   d_thr  [2][4][P] float : the value AmpliSolveVariantCalling reads back from the table text:
 *                            stof(sprintf("%f", rate)) (EE:1704 -> VC:889-890), 0.01f for code != 0
 *                            (EE:2680-2684); computed in exact integer arithmetic on the device
 *   d_germ_val [4][P] float, d_germ_present [4][P] uint8 : Germ_Max cell (EE:2807-2849)
 *   d_flags [1] int32 : bit 0 set when a sum left the exactness envelope
 * Any output pointer except d_code/d_rate may be NULL.
 */
int ampli_error_finalize(ampli_ctx *ctx, const ampli_acc_table *d_acc, float C,
                         int32_t coverage_cutoff, float *d_rate, uint8_t *d_code, float *d_thr,
                         float *d_germ_val, uint8_t *d_germ_present, int32_t *d_flags);

/* one emitted call (VC:898 true): everything the post-call annotation needs of the record, so that the host does
 * not have to keep the record array once it is uploaded */
typedef struct ampli_call {
    int32_t sample;   /* index into the T axis of this launch */
    int32_t record;   /* r in [0, R) */
    int32_t alt;      /* 0..3 = A,C,G,T */
    int32_t rd;       /* the RD column of the line (== fw + bw unless the line was irregular, VC:762-765) */
    double q_fw, q_bw;            /* VC:895-896 */
    float af, af_fw, af_bw;       /* VC:772-817: the reported VAFs */
    int32_t k_fw, k_bw;           /* alt reads per strand (Xfw = X - Xrs, Xrs) */
    int32_t fw, bw;               /* strand depths: sums of the four forward / reverse counts (VC:760-761) */
    int32_t flags;                /* AMPLI_CALL_* */
} ampli_call; /* 64 bytes */

/* ampli_call.flags.  The device forms Q = -10 log10 p in fp64 with ROCm's exp / log, the reference in glibc's double and
 * x87 long double (VC:3866-3880): the two agree to ~1e-10, so a Q within 1e-6 of the gate Q >= 5 (VC:898) cannot be decided on
 * the device.  Such (record, alternative) pairs are put on the call list EITHER WAY with AMPLI_CALL_BORDERLINE set; the mask
 * bit follows the device's own value, and the host re-evaluates the pair with the reference's operation sequence before it
 * writes anything (csrc/host/annotate.cpp: score_reference_sequence).  None occurs on Toy_data or the synthetic panels. */
#define AMPLI_CALL_BORDERLINE 1
#define AMPLI_CALL_GATE_EPS 1e-6

#define AMPLI_CALL_SHARDS 32
#define AMPLI_CALL_COUNTER_STRIDE 16 /* uint64 words between shard counters: one 128-byte line each */
#define AMPLI_CALL_COUNTER_WORDS (AMPLI_CALL_SHARDS * AMPLI_CALL_COUNTER_STRIDE)

#define AMPLI_POISSON_FULL 0      /* evaluate all 6 scores of every record, as the reference does */
#define AMPLI_POISSON_PREFILTER 1 /* skip scores an exact bound proves < 5 (identical outputs) */

/*
 * poisson_call -- replaces the per-line core of callVariants (VC:752-898 and
 * its 11 clones) with mutationRulesPoissonQualityScore / kf_gammaq / kf_lgamma
 * (VC:3721-3884) for T tumour samples.
 *   d_thr [2][4][P] as produced by error_finalize or parsed from the table (std::stof, VC:889-890)
 *   d_ref_code [P]  0..3 = A,C,G,T; 255 = reference base not in ACGT -> record skipped (VC:3290)
 *   d_call_mask [T][R] uint8: bit a set = alt nucleotide a called at that record (4-byte aligned buffer,
 *     rounded up to a multiple of 4 bytes)
 *   d_calls / capacity / d_n_calls: optional compact list of emitted calls, kept as AMPLI_CALL_SHARDS independent
 *     segments so that appends do not serialise on one counter: segment k = entries [k*(capacity/SHARDS), ...),
 *     its fill count is d_n_calls[k*AMPLI_CALL_COUNTER_STRIDE] (a count above capacity/SHARDS means that segment
 *     overflowed: rerun with a larger capacity).  d_n_calls points to AMPLI_CALL_COUNTER_WORDS uint64 words; the
 *     call resets them itself.  Entries are unordered: sort by (sample, record, alt) for the reference's
 *     emission order.  d_n_calls without d_calls just counts.
 *   d_q [T][R][4][2] double, optional dense scores (Q_fw,Q_bw per nucleotide; -1 = not evaluated;
 *     requires mode FULL); d_af [T][R][4][3] float optional dense {AF, AF_fw, AF_bw}.
 */
int ampli_poisson_call(ampli_ctx *ctx, const int32_t *d_trecs, int64_t P, int64_t E,
                       const uint32_t *d_ext_pos, int32_t T, const float *d_thr,
                       const uint8_t *d_ref_code, int32_t coverage_cutoff, int32_t mode,
                       uint8_t *d_call_mask, ampli_call *d_calls, int64_t capacity,
                       unsigned long long *d_n_calls, double *d_q, float *d_af);
/* the same with the thresholds read straight from the all-gathered blocks of a position-sliced merge (d_blocks as
 * ampli_error_table_unslice takes them): spares the unslice launch when only the calls are wanted */
int ampli_poisson_call_blocks(ampli_ctx *ctx, const int32_t *d_trecs, int64_t P, int64_t E, const uint32_t *d_ext_pos,
                              int32_t T, const void *d_blocks, int32_t n_slices, const uint8_t *d_ref_code,
                              int32_t coverage_cutoff, int32_t mode, uint8_t *d_call_mask, ampli_call *d_calls,
                              int64_t capacity, unsigned long long *d_n_calls, double *d_q, float *d_af);

/* poisson_call over an explicitly described cohort / chunk (ampli_records above); d_call_mask is [n_samples][P + E] */
int ampli_poisson_call_records(ampli_ctx *ctx, const ampli_records *trecs, int64_t P, const float *d_thr,
                               const uint8_t *d_ref_code, int32_t coverage_cutoff, int32_t mode, uint8_t *d_call_mask,
                               ampli_call *d_calls, int64_t capacity, unsigned long long *d_n_calls, double *d_q, float *d_af);

/* Asynchronous drain (opt-in).  In prefilter mode poisson_call is two kernels; the second (the dense drain of the
 * queued survivors) is one fp64 scorer chain long and independent of what the caller enqueues next.  With
 * ampli_set_async_drain(ctx, 1) it runs on a side stream of the context: the call mask, the call list and
 * d_n_calls of a poisson_call are then complete only after ampli_wait_calls (the context's stream waits, the host
 * does not block), ampli_sync, ampli_copy_d2h, ampli_ctx_flags or the next ampli_poisson_call; the outputs must
 * stay untouched until then (the queued items carry their own thresholds and depths). */
int ampli_set_async_drain(ampli_ctx *ctx, int32_t on);
int ampli_wait_calls(ampli_ctx *ctx);

/* hipGraph capture of a sequence of calls on the context's stream (small, launch-bound panels).  The context must
 * own a real stream (AMPLI_STREAM_OWN or a non-null stream) and the sequence must have run once (workspaces warm).
 * ampli_graph_end returns a hipGraphExec_t as an opaque pointer; launch it any number of times. */
int ampli_graph_begin(ampli_ctx *ctx);
int ampli_graph_end(ampli_ctx *ctx, void **graph_exec);
int ampli_graph_launch(ampli_ctx *ctx, void *graph_exec);
int ampli_graph_destroy(void *graph_exec);

/*
 * Native transport of the multi-GPU merge (one process per GPU, no Python): RCCL over xGMI, bound at run time
 * (librccl.so is only loaded when a communicator is created).  Rendezvous through a file every rank can see: rank 0
 * removes whatever file of that name a dead run left behind and writes its ncclUniqueId there, tagged with the world size,
 * the launch's AMPLISOLVE_JOB_NONCE (optional; any integer the launcher gives every rank) and the time; the others wait up
 * to timeout_s for a file that carries THEIR world size and nonce and is not older than the job, and ncclCommInitRank
 * itself is bounded by timeout_s as well (RCCL has no timeout of its own): a missing rank or a stale file ends in
 * AMPLI_E_HIP (no usable id file) or AMPLI_E_COMM_TIMEOUT (the init itself timed out: the caller must _exit, see the
 * code's comment) with a message, never in a hang.  A rank above 0 reads the file once more 0.25 s after accepting it and
 * takes the newer one, so a relaunch right after a crash does not pick up the dead run's id in the moment before rank 0
 * replaces it (without a nonce that window cannot be closed completely: give every launch its AMPLISOLVE_JOB_NONCE).  Collectives are enqueued on the context's
 * stream; the *_i32 / *_i64 helpers take HOST values and synchronise.  Buffer shapes as in "Position-sliced merge".
 */
typedef struct ampli_comm ampli_comm;
int ampli_comm_create(ampli_ctx *ctx, int32_t rank, int32_t world, const char *id_file, int32_t timeout_s, ampli_comm **out);
void ampli_comm_destroy(ampli_comm *c);
int ampli_comm_reduce_scatter_f64(ampli_comm *c, const double *d_send /*[world][count]*/, double *d_recv /*[count]*/, int64_t count);
int ampli_comm_all_to_all_f32(ampli_comm *c, const float *d_send /*[world][count]*/, float *d_recv /*[world][count]*/, int64_t count);
int ampli_comm_all_gather_bytes(ampli_comm *c, const void *d_send /*[bytes]*/, void *d_recv /*[world][bytes]*/, int64_t bytes);
int ampli_comm_all_reduce_max_i32(ampli_comm *c, int32_t *values /*host, in place*/, int32_t n);
int ampli_comm_exclusive_sum_i64(ampli_comm *c, int64_t mine, int64_t *before);
int ampli_comm_barrier(ampli_comm *c);

/* scalar scorer on the device for known-answer tests: q[i] = score(k[i], rd[i], err[i]),
 * p[i] = 1 - kf_gammaq(k, rd*err) (VC:3834-3884).  Either output may be NULL. */
int ampli_score_batch(ampli_ctx *ctx, const int32_t *d_k, const int32_t *d_rd, const float *d_err,
                      int64_t n, double *d_q, double *d_p);
/* the scorer of the all-scores mode (AMPLI_POISSON_FULL) for the same tests: the integer-count form of the same recipe --
 * lgamma from a table of the Lanczos form's own values, division-free series and continued fraction (csrc/ampli_math.h,
 * ampli_poisson_score_dense); q[i] equals ampli_score_batch's to rounding (~1e-13 relative on p) */
int ampli_score_dense_batch(ampli_ctx *ctx, const int32_t *d_k, const int32_t *d_rd, const float *d_err, int64_t n, double *d_q);
/* text round trip on the device for known-answer tests: out[i] = stof(sprintf("%f", in[i])) */
int ampli_roundtrip_batch(ampli_ctx *ctx, const float *d_in, int64_t n, float *d_out);

/* deterministic synthetic panel generated in HBM: fills recs[n_samples][P][8]
 * for samples [first_sample, first_sample+n_samples) of panel `seed`; tumour != 0 spikes SNVs.
 * Bit-identical to ampli_synth_record() on the host (amplisolve_amd/csrc/ampli_synth.h). */
int ampli_synth_fill(ampli_ctx *ctx, int32_t *d_recs, int64_t P, int32_t n_samples,
                     int32_t first_sample, uint64_t seed, int32_t depth, int32_t tumour);
int ampli_synth_ref(ampli_ctx *ctx, uint8_t *d_ref_code, int64_t P, uint64_t seed);

/*
 * Upstream of the path: alignments -> per-position counts, i.e. what ASEQ's PILEUP mode / the reference's binary-only
 * computeCounts produce (/root/reference/Execution_examples.md:16-46: vcf= bam= mbq= mrq= mdc=).  d_bam: the UNCOMPRESSED BAM
 * alignment records (the host inflates the BGZF blocks); d_rec_off[i]: byte offset of record i's block_size field, every listed
 * record lying completely inside d_bam with a CIGAR that matches its l_seq (the host checks), and the buffer readable for 16 bytes
 * behind the last record (the kernel copies whole 16-byte pieces; d_bam itself 16-byte aligned); d_keys[P]: the panel's unique
 * positions as (BAM reference id << 32 | 1-based position), ascending.  d_counts int32 [P][8] = {A,C,G,T, Ars,Crs,Grs,Trs} is
 * ACCUMULATED into (zero it first; batches of one file add up).  Kept reads: mapped, not secondary / QC-fail / duplicate,
 * MAPQ >= mrq; counted bases: M / = / X columns, A/C/G/T, base quality >= mbq.  d_stats (optional, 2 words, accumulated):
 * reads kept, bases counted.
 */
int ampli_pileup_count(ampli_ctx *ctx, const uint8_t *d_bam, const uint64_t *d_rec_off, int64_t n_reads, const uint64_t *d_keys, int64_t P,
                       int32_t mbq, int32_t mrq, int32_t *d_counts, uint64_t *d_stats);

/* tuning knobs.  reduce_sample_splits: 0 = automatic.  reduce_general: 0 = the fast error_reduce kernel
 * (valid while every strand depth is < 2^22; it raises AMPLI_FLAG_RERUN_GENERAL otherwise), 1 = the literal
 * kernel that follows the reference operation by operation for any depth (slower).  reduce_lane_groups: lane
 * groups per wave of error_reduce (1, 2 or 4; 0 = automatic: 1 unless the panel is too small to fill the chip). */
int ampli_set_tuning(ampli_ctx *ctx, int32_t reduce_sample_splits, int32_t reduce_general, int32_t reduce_lane_groups);
/* error_reduce with a compact per-position state (96 VGPRs: five waves per SIMD instead of four) -- error_reduce_u16_kernel for
 * uint16 records (at most 4096 samples per launch) and, since round 5, error_reduce_u24_kernel for 24-bit records (at most 4092
 * samples; a covered record with RD >= 2^22 raises AMPLI_FLAG_RERUN_GENERAL like the fast general kernel): taken by
 * ampli_error_estimate / ampli_error_reduce_records(_sliced) / ampli_error_reduce_sliced when on != 0 (the default) and the launch
 * has the shape they cover -- fast kernel, one lane group, one sample split, and either no accumulator table (finalize fused, or
 * the shard's sums straight into the sliced exchange buffers) or a table the caller takes as streaming state
 * (AMPLI_REDUCE_SUMMARY).  Positions listed more than once (E > 0) are served: their tiles go to the general kernel over a list.
 * Every other launch takes the general kernel.  Same results, bit for bit.  on = 2: the uint16 form only (A/B runs). */
int ampli_set_reduce_compact(ampli_ctx *ctx, int32_t on);
/* Which kernel the context's latest error_reduce launch was: 0 = error_reduce_kernel (general), 1 = error_reduce_u16_kernel,
 * 2 = error_reduce_u24_kernel (compact state); AMPLI_E_INVALID before the first launch.  For tests and the bench line, which name the kernel they measured. */
int ampli_last_reduce_kernel(const ampli_ctx *ctx);

/*
 * Position ranges on concurrent streams (round 5).  A resident panel is rarely a whole number of rounds of workgroups (config 3:
 * 1563 tiles of 64 positions on 1280 resident workgroups of error_reduce), and a launch's partly filled last round runs at a
 * fraction of the chip.  With n_ranges > 1 (at most 4) ampli_error_estimate / ampli_error_reduce_records and ampli_poisson_call /
 * _records (prefilter mode) cut the panel into n tile-aligned ranges of positions, each on a stream the context owns (the
 * context's own stream only forks into them and joins them), each range's poisson_call behind its own error_estimate (a position's
 * thresholds are all a record of that position needs).  Over BACK-TO-BACK passes on independent batches one range's poisson_call
 * and another's error_reduce then fill each other's thin rounds (config 3: 0.143-0.155 -> 0.132-0.139 ms per pass; more than two
 * ranges buy nothing more); a single pass gains nothing -- the
 * join at its end costs what the overlap inside it saves -- so the command lines do not use it.  Outputs are the same arrays,
 * bit for bit; the call list's shards are dealt to the ranges (range k appends to shards [32 k / n, 32 (k + 1) / n)), so a shard
 * fills n times faster than without ranges.
 * Semantics: the section opens at the first such call (the ranges' streams wait for everything enqueued on the context's stream
 * so far) and CLOSES -- the context's stream waits for every range -- at the next call of any other entry point on the context
 * (ampli_sync, copies, ampli_event_record, ampli_ctx_flags, every other kernel, ampli_stream), or at ampli_ranges_join.  While
 * it is open the caller must not enqueue work of its own that writes the ranges' inputs or reads their outputs.  Launches outside
 * the shape (positions listed more than once, another record layout than the compact kernel's for error_estimate, the all-scores
 * mode, a panel of fewer than 2 n tiles, P + E not a multiple of 4 for poisson_call, stream capture) simply run unsplit, after a join.
 */
int ampli_set_ranges(ampli_ctx *ctx, int32_t n_ranges);
int ampli_ranges_join(ampli_ctx *ctx);
/* 1 when ampli_set_ranges saw every pair of the ranges' streams run concurrently; 0 when it could not find streams on different
 * hardware queues (HIP deals streams to a few queues by rules of its own: ampli_set_ranges probes each new stream against the earlier
 * ranges' with a short sleeping kernel and replaces one that shares a queue) -- results are the same either way, the overlap is not. */
int ampli_ranges_concurrent(const ampli_ctx *ctx);
/* An event (ampli_event_create) on range `range`'s stream WITHOUT closing the section: events recorded before and after a call
 * bracket that range's share of it -- the kernels' durations under the overlap the ranges exist for.  (ampli_event_record is an
 * ordinary call: it closes the section first and records on the context's stream.) */
int ampli_range_event_record(ampli_ctx *ctx, int32_t range, void *ev);

/* poisson_call (prefilter mode) launch shape; 0 = default for each.  rows_per_wave: tumour rows one wave streams
 * (a workgroup = 4 waves over one 64-record tile and 4 * rows_per_wave rows).  drain_blocks_per_shard: workgroups per
 * queue shard in the drain launch.  Results do not depend on either. */
int ampli_set_poisson_tuning(ampli_ctx *ctx, int32_t rows_per_wave, int32_t drain_blocks_per_shard);

/* Minimum capacity (items of 40 B) of the prefilter queue of ampli_poisson_call; default T*R/4, at least 65536. */
int ampli_set_queue_items(ampli_ctx *ctx, int64_t items);

/* Flags raised by kernels of this context since the last clear (synchronises the stream). */
#define AMPLI_FLAG_QUEUE_OVERFLOW 4 /* poisson_call (prefilter) ran out of queue space: masks/calls incomplete, raise
                                       ampli_set_queue_items (or use AMPLI_POISSON_FULL) and rerun */
#define AMPLI_FLAG_SLICE_RANGE 8 /* sliced exchange in the slim format: a shard's value does not fit its share of a packed field; repeat
                                  * the exchange with ampli_set_slice_format(ctx, AMPLI_SLICE_WIDE) */
#define AMPLI_FLAG_RERUN_GENERAL 2 /* error_reduce met a depth >= 2^22: its table is invalid, rerun with reduce_general = 1 */
int ampli_ctx_flags(ampli_ctx *ctx, int32_t *out, int32_t clear);

#ifdef __cplusplus
}
#endif
#endif /* AMPLISOLVE_HIP_H */
