/*
 * oracle/ampli_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded CPU restatement of AmpliSolve's per-position error
 * estimation + Poisson calling hot path.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may link or call this.  The shipped path
 * (amplisolve_amd/csrc) never does.
 *
 * Citation legend (reference is read-only at /root/reference):
 *   EE:n = source_codes/AmpliSolveErrorEstimation.cpp:n
 *   VC:n = source_codes/AmpliSolveVariantCalling.cpp:n
 *
 * Pinning status:
 *   - error-estimation half: pinned against the compiled reference itself
 *     (oracle/_ref/ee_ref_driver, built from EE.cpp where it lies) on Toy_data
 *     and synthetic minis -> tests/golden/ee_*.
 *   - Poisson scorer (kf_lgamma/kf_gammaq/score): pinned against the scorer
 *     functions of VC.cpp compiled from the source where it lies
 *     (oracle/_ref/libvc_scorer_ref.so) and against the known answers the
 *     survey captured -> tests/golden/vc_scorer_*.
 *   - callVariants' per-line gate (VC:723-898): the whole VC translation unit
 *     is UNBUILDABLE here (needs Boost.Math, absent; no stand-in is written),
 *     so the gate restatement is pinned only through the scorer + the survey's
 *     recorded Toy_data call rows.
 *
 * Data layout shared with the product (see include/amplisolve_hip.h):
 *   records  int32 recs[n_samples][R][8], R = P + E
 *            fields {Afw,Cfw,Gfw,Tfw,Ars,Crs,Grs,Trs}; Xfw = X - Xrs (EE:1155-1158)
 *            absent record: recs[..][0] == INT32_MIN
 *   P        unique panel positions; record r < P is the first occurrence of
 *            position r in the sample's file
 *   E        extra occurrences (a position listed again in the same file,
 *            overlapping amplicons); extras of position p are records
 *            P + dup_off[p] .. P + dup_off[p+1]-1, visited right after (s,p)
 */
#ifndef AMPLI_ORACLE_H
#define AMPLI_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORACLE_ABSENT INT32_MIN

/* ---- Poisson scorer (VC:3721-3884) ---- */
double oracle_kf_lgamma(double z);                 /* VC:3817-3830 */
double oracle_kf_gammap_series(double s, double z);/* VC:3785-3794 (_kf_gammap) */
double oracle_kf_gammaq_cf(double s, double z);    /* VC:3733-3752 (_kf_gammaq) */
double oracle_kf_gammaq(double s, double z);       /* VC:3726-3729 */
long double oracle_score(int k, int RD, float err);/* VC:3834-3884 */
/* p-value as the reference forms it before the clamp: 1 - kf_gammaq(k, RD*err); 1 when k==0 */
double oracle_pvalue(int k, int RD, float err);
void oracle_score_batch(const int32_t *k, const int32_t *rd, const float *err, int64_t n,
                        double *q_out, double *p_out);

/* ---- error estimation (EE:1057-1481, EE:1484-2544) ---- */
void oracle_error_reduce(const int32_t *recs, int64_t P, int64_t E, const uint32_t *dup_off,
                         int32_t S, int32_t first_sample, float C, int32_t cov,
                         double *snt /*[2][4][P]*/, int64_t *srd /*[2][4][P]*/,
                         int32_t *cnt /*[4][P]*/, int32_t *nrec /*[P]*/,
                         int32_t *gm_n /*[4][P]*/, int32_t *gm_first /*[4][P]*/,
                         float *gm_first_af /*[4][P]*/, float *gm_rest /*[4][P]*/,
                         int32_t *order_sensitive /* out: 1 if reverse-order double sums differ */);
/* the same with the RD column of every record (EE:1149): rdcol [S][P+E], INT32_MIN where it equals A+C+G+T; NULL = all
 * regular.  A line whose RD differs is used with its own RD in the Germ_Max AF (EE:1229-1232). */
void oracle_error_reduce_rd(const int32_t *recs, const int32_t *rdcol, int64_t P, int64_t E, const uint32_t *dup_off,
                            int32_t S, int32_t first_sample, float C, int32_t cov, double *snt, int64_t *srd, int32_t *cnt,
                            int32_t *nrec, int32_t *gm_n, int32_t *gm_first, float *gm_first_af, float *gm_rest,
                            int32_t *order_sensitive);

/* snt [2][4][P] in the reference's own order of addition (reverse insertion order of its multimap: last sample first, a position's
 * later lines before its first); equals oracle_error_reduce's snt inside the exactness envelope, IS the reference's double outside it */
void oracle_error_sums_inorder(const int32_t *recs, const int32_t *rdcol, int64_t P, int64_t E, const uint32_t *dup_off,
                               int32_t S, float C, int32_t cov, double *snt);

/* ordered merge of two partial tables (L = earlier samples); result into L */
void oracle_acc_merge(int64_t P, double *snt, int64_t *srd, int32_t *cnt, int32_t *nrec,
                      int32_t *gm_n, int32_t *gm_first, float *gm_first_af, float *gm_rest,
                      const double *snt_r, const int64_t *srd_r, const int32_t *cnt_r,
                      const int32_t *nrec_r, const int32_t *gm_n_r, const int32_t *gm_first_r,
                      const float *gm_first_af_r, const float *gm_rest_r);

/* code: 0 = estimate present, 1 = below quorum (EE:1659), 2 = NaN (EE:1682) */
void oracle_error_finalize(int64_t P, const double *snt, const int64_t *srd, const int32_t *cnt,
                           const int32_t *nrec, const int32_t *gm_n, const float *gm_rest,
                           float *rate /*[2][4][P]*/, uint8_t *code /*[4][P]*/,
                           float *thr /*[2][4][P] value AmpliSolveVariantCalling would read back*/,
                           double *germ_val /*[4][P]*/, uint8_t *germ_present /*[4][P]*/);

/* the threshold cell text (EE:1704, EE:2670-2688); returns length */
int oracle_format_thr_cell(uint8_t code, float r_fw, float r_bw, int is_ref, char *buf);
/* "%f" -> std::stof round trip of one rate (EE:1704 -> VC:889) */
float oracle_text_roundtrip(float r);
/* germ-max cell text: "-" or ostream<<double (EE:2807-2849) */
int oracle_format_germ_cell(uint8_t present, double v, char *buf);

/* ---- per tumour record evaluation (VC:723-898) ---- */
void oracle_poisson_call(const int32_t *trecs, int64_t P, int64_t E, const uint32_t *ext_pos,
                         int32_t T, const float *thr /*[2][4][P]*/, const uint8_t *ref_code /*[P]*/,
                         int32_t cov, uint8_t *call_mask /*[T][R]*/,
                         double *q /*optional [T][R][4][2], -1 = not evaluated*/,
                         float *af /*optional [T][R][4][3] = AF, AF_fw, AF_bw*/);
/* the same with the RD column of every record (VC:752): forward depth RD - RD_reverse (VC:895), AF = X / RD (VC:814-817) */
void oracle_poisson_call_rd(const int32_t *trecs, const int32_t *rdcol, int64_t P, int64_t E, const uint32_t *ext_pos,
                            int32_t T, const float *thr, const uint8_t *ref_code, int32_t cov, uint8_t *call_mask, double *q,
                            float *af);

/* ---- integer AF-gate equivalence helper used by the product kernels ---- */
/* 1 iff (double)((float)x/(float)d) <= 0.05, evaluated exactly like EE:1592-1595 */
int oracle_af_gate(int32_t x, int32_t d);
void oracle_af_gate_batch(const int32_t *x, const int32_t *d, int64_t n, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
