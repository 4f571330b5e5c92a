/*
 * oracle/ampli_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see ampli_oracle.h).
 *
 * CPU restatement of the AmpliSolve hot path on position-indexed arrays.  The
 * reference keeps everything in string-keyed hash maps; the arithmetic below
 * is the arithmetic those ~6900 lines perform, in the same types and the same
 * operation order, each block citing the lines it follows.
 *
 * Build: gcc -O2 -ffp-contract=off (plain x86-64: no FMA contraction, so
 * `sum + base + float*float` evaluates exactly as the reference's does).
 */
#include "ampli_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define KF_GAMMA_EPS 1e-14 /* VC:149 */
#define KF_TINY 1e-290     /* VC:150 */

/* ------------------------------------------------------------------ */
/* Poisson scorer                                                      */
/* ------------------------------------------------------------------ */

/* VC:3817-3830: 8-term Lanczos log-gamma */
double oracle_kf_lgamma(double z)
{
    double x = 0;
    x += 0.1659470187408462e-06 / (z + 7);
    x += 0.9934937113930748e-05 / (z + 6);
    x -= 0.1385710331296526 / (z + 5);
    x += 12.50734324009056 / (z + 4);
    x -= 176.6150291498386 / (z + 3);
    x += 771.3234287757674 / (z + 2);
    x -= 1259.139216722289 / (z + 1);
    x += 676.5203681218835 / z;
    x += 0.9999999999995183;
    return log(x) - 5.58106146679532777 - z + (z - 0.5) * log(z + 6.5);
}

/* VC:3785-3794: regularized lower incomplete gamma by series, <=99 terms */
double oracle_kf_gammap_series(double s, double z)
{
    double sum, x;
    int k;
    for (k = 1, sum = x = 1.; k < 100; ++k) {
        sum += (x *= z / (s + k));
        if (x / sum < KF_GAMMA_EPS) break;
    }
    return exp(s * log(z) - z - oracle_kf_lgamma(s + 1.) + log(sum));
}

/* VC:3733-3752: regularized upper incomplete gamma by modified Lentz, <=99 steps */
double oracle_kf_gammaq_cf(double s, double z)
{
    int j;
    double C, D, f;
    f = 1. + z - s;
    C = f;
    D = 0.;
    for (j = 1; j < 100; ++j) {
        double a = j * (s - j), b = (j << 1) + 1 + z - s, d;
        D = b + a * D;
        if (D < KF_TINY) D = KF_TINY;
        C = b + a / C;
        if (C < KF_TINY) C = KF_TINY;
        D = 1. / D;
        d = C * D;
        f *= d;
        if (fabs(d - 1.) < KF_GAMMA_EPS) break;
    }
    return exp(s * log(z) - z - oracle_kf_lgamma(s) - log(f));
}

/* VC:3726-3729 */
double oracle_kf_gammaq(double s, double z)
{
    return z <= 1. || z < s ? 1. - oracle_kf_gammap_series(s, z) : oracle_kf_gammaq_cf(s, z);
}

/* VC:3864-3865 (and VC:3858-3861 for k == 0, VC:3852-3856 for err == 0) */
double oracle_pvalue(int k, int RD, float err)
{
    if (err == 0) err = 0.0010008;
    if (k == 0) return 1.0;
    double m = (double)RD * err;
    return 1 - oracle_kf_gammaq(k, m);
}

/* VC:3834-3884 */
long double oracle_score(int k, int RD, float err)
{
    long double Q = 0;
    long double pvalue = 0;
    long double p_limit = 0.0000000001;
    long double m = 0;
    if (err == -1) {
        Q = -888;
        return Q;
    }
    if (err == 0) err = 0.0010008;
    if (k == 0) {
        pvalue = 1;
    } else {
        m = (double)RD * err;
        pvalue = 1 - oracle_kf_gammaq(k, m);
    }
    if (pvalue < p_limit) {
        pvalue = p_limit;
        Q = -10 * log10l(pvalue);
    } else if (pvalue == 1) {
        Q = 0;
    } else {
        Q = -10 * log10l(pvalue);
    }
    return Q;
}

void oracle_score_batch(const int32_t *k, const int32_t *rd, const float *err, int64_t n,
                        double *q_out, double *p_out)
{
    for (int64_t i = 0; i < n; ++i) {
        if (q_out) q_out[i] = (double)oracle_score(k[i], rd[i], err[i]);
        if (p_out) p_out[i] = (err[i] == -1) ? -1.0 : oracle_pvalue(k[i], rd[i], err[i]);
    }
}

/* ------------------------------------------------------------------ */
/* Error estimation                                                    */
/* ------------------------------------------------------------------ */

/* EE:1592-1595: AF computed in fp32, widened, compared with the double 0.05 */
int oracle_af_gate(int32_t x, int32_t d)
{
    double af = (float)x / (float)d;
    return af <= 0.05;
}

void oracle_af_gate_batch(const int32_t *x, const int32_t *d, int64_t n, uint8_t *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = (uint8_t)oracle_af_gate(x[i], d[i]);
}

/* rd_col: the RD column of the line, or ORACLE_ABSENT for "equal to A+C+G+T" (the usual case; a line where it differs
 * prints EE:1178-1181's message and is then used with its own RD: EE:1229-1232 divide by it) */
static inline void visit_record(const int32_t *r, int32_t rd_col, int32_t sample_index, float C, int32_t cov,
                                int64_t p, int64_t P, double *snt, double *srd_d, int32_t *cnt,
                                int32_t *nrec, int32_t *gm_n, int32_t *gm_first,
                                float *gm_first_af, float *gm_rest)
{
    if (r[0] == ORACLE_ABSENT) return;
    /* EE:1175-1176 */
    int FW = r[0] + r[1] + r[2] + r[3];
    int BW = r[4] + r[5] + r[6] + r[7];
    int RD = rd_col == ORACLE_ABSENT ? FW + BW : rd_col; /* the ASEQ RD column (EE:1149) */
    nrec[p] += 1;     /* Value_Hash.count(key), EE:1659 */
    for (int nt = 0; nt < 4; ++nt) {
        int base_fw = r[nt], base_bw = r[4 + nt];
        /* EE:1592-1606 */
        double AF_limit = 0.05;
        double AF_fw = (float)base_fw / (float)FW;
        double AF_bw = (float)base_bw / (float)BW;
        if (AF_fw <= AF_limit && AF_bw <= AF_limit && FW >= cov && BW >= cov) {
            double *sf = &snt[(0 * 4 + nt) * P + p], *sb = &snt[(1 * 4 + nt) * P + p];
            *sf = *sf + base_fw + ((float)FW * (float)C);
            srd_d[(0 * 4 + nt) * P + p] += FW;
            *sb = *sb + base_bw + ((float)BW * (float)C);
            srd_d[(1 * 4 + nt) * P + p] += BW;
            cnt[nt * P + p] += 1;
        }
        /* EE:1229-1232, EE:1251-1271 (A) and clones; sentinel handled at finalize */
        float AF = (float)(base_fw + base_bw) / (float)RD;
        if (AF <= 0.05 && FW >= cov && BW >= cov) {
            int64_t i = nt * P + p;
            if (gm_n[i] == 0) {
                gm_first[i] = sample_index;
                gm_first_af[i] = AF;
            } else if (gm_rest[i] <= AF) {
                gm_rest[i] = AF;
            }
            gm_n[i] += 1;
        }
    }
}

void oracle_error_reduce(const int32_t *recs, int64_t P, int64_t E, const uint32_t *dup_off,
                         int32_t S, int32_t first_sample, float C, int32_t cov, double *snt,
                         int64_t *srd, int32_t *cnt, int32_t *nrec, int32_t *gm_n,
                         int32_t *gm_first, float *gm_first_af, float *gm_rest,
                         int32_t *order_sensitive)
{
    oracle_error_reduce_rd(recs, NULL, P, E, dup_off, S, first_sample, C, cov, snt, srd, cnt, nrec, gm_n, gm_first,
                           gm_first_af, gm_rest, order_sensitive);
}

/* the same with the RD column of every record: rdcol [S][P+E], ORACLE_ABSENT where it equals A+C+G+T; NULL = all regular */
void oracle_error_reduce_rd(const int32_t *recs, const int32_t *rdcol, int64_t P, int64_t E, const uint32_t *dup_off,
                            int32_t S, int32_t first_sample, float C, int32_t cov, double *snt,
                            int64_t *srd, int32_t *cnt, int32_t *nrec, int32_t *gm_n,
                            int32_t *gm_first, float *gm_first_af, float *gm_rest,
                            int32_t *order_sensitive)
{
    const int64_t R = P + E;
#define RDC(s_, r_) (rdcol ? rdcol[(size_t)(s_) * R + (r_)] : ORACLE_ABSENT)
    double *srd_d = (double *)calloc((size_t)(8 * P), sizeof(double));
    for (int64_t i = 0; i < 8 * P; ++i) snt[i] = 0.0;
    for (int64_t i = 0; i < 4 * P; ++i) {
        cnt[i] = 0;
        gm_n[i] = 0;
        gm_first[i] = INT32_MAX;
        gm_first_af[i] = 0.0f;
        gm_rest[i] = -INFINITY;
    }
    for (int64_t p = 0; p < P; ++p) nrec[p] = 0;

    /* visit order: samples in list order, lines in file order (EE:1081, EE:1114) */
    for (int64_t p = 0; p < P; ++p) {
        for (int32_t s = 0; s < S; ++s) {
            const int32_t *base = recs + (size_t)s * R * 8;
            visit_record(base + p * 8, RDC(s, p), first_sample + s, C, cov, p, P, snt, srd_d, cnt, nrec, gm_n,
                         gm_first, gm_first_af, gm_rest);
            if (dup_off)
                for (uint32_t e = dup_off[p]; e < dup_off[p + 1]; ++e)
                    visit_record(base + (P + e) * 8, RDC(s, P + e), first_sample + s, C, cov, p, P, snt, srd_d,
                                 cnt, nrec, gm_n, gm_first, gm_first_af, gm_rest);
        }
    }
    for (int64_t i = 0; i < 8 * P; ++i) srd[i] = (int64_t)srd_d[i];

    if (order_sensitive) {
        /* libstdc++'s equal_range walks equal keys in reverse insertion order; the
           reference's double sums are order-free only inside the exactness envelope
           (DESIGN.md).  Re-sum backwards and report any difference. */
        *order_sensitive = 0;
        double *snt2 = (double *)calloc((size_t)(8 * P), sizeof(double));
        double *srd2 = (double *)calloc((size_t)(8 * P), sizeof(double));
        int32_t *cnt2 = (int32_t *)calloc((size_t)(4 * P), sizeof(int32_t));
        int32_t *nrec2 = (int32_t *)calloc((size_t)P, sizeof(int32_t));
        int32_t *g1 = (int32_t *)calloc((size_t)(4 * P), sizeof(int32_t));
        int32_t *g2 = (int32_t *)calloc((size_t)(4 * P), sizeof(int32_t));
        float *g3 = (float *)calloc((size_t)(4 * P), sizeof(float));
        float *g4 = (float *)calloc((size_t)(4 * P), sizeof(float));
        for (int64_t p = 0; p < P; ++p) {
            for (int32_t s = S - 1; s >= 0; --s) {
                const int32_t *base = recs + (size_t)s * R * 8;
                if (dup_off)
                    for (uint32_t e = dup_off[p + 1]; e > dup_off[p]; --e)
                        visit_record(base + (P + e - 1) * 8, RDC(s, P + e - 1), s, C, cov, p, P, snt2, srd2, cnt2,
                                     nrec2, g1, g2, g3, g4);
                visit_record(base + p * 8, RDC(s, p), s, C, cov, p, P, snt2, srd2, cnt2, nrec2, g1, g2, g3, g4);
            }
        }
        if (memcmp(snt, snt2, (size_t)(8 * P) * sizeof(double)) != 0) *order_sensitive = 1;
        free(snt2); free(srd2); free(cnt2); free(nrec2); free(g1); free(g2); free(g3); free(g4);
    }
    free(srd_d);
#undef RDC
}

/* The eight threshold sums in the order the reference adds them: estimateThresholds walks `equal_range` of an
 * unordered_multimap (EE:1555, 1565-1606) that storeGermlineStatistics filled in visit order (EE:1245 and clones), and libstdc++
 * hands equal keys back in REVERSE insertion order -- the last file first, within a file a position's later lines before its
 * first.  Inside the exactness envelope this equals oracle_error_reduce's snt; outside it, this is the reference's double
 * (pinned against the compiled reference on cohorts outside the envelope: tests/test_oracle_golden.py). */
void oracle_error_sums_inorder(const int32_t *recs, const int32_t *rdcol, int64_t P, int64_t E, const uint32_t *dup_off,
                               int32_t S, float C, int32_t cov, double *snt)
{
    const int64_t R = P + E;
#define RDC(s_, r_) (rdcol ? rdcol[(size_t)(s_) * R + (r_)] : ORACLE_ABSENT)
    double *srd2 = (double *)calloc((size_t)(8 * P), sizeof(double));
    int32_t *cnt2 = (int32_t *)calloc((size_t)(4 * P), sizeof(int32_t));
    int32_t *nrec2 = (int32_t *)calloc((size_t)P, sizeof(int32_t));
    int32_t *g1 = (int32_t *)calloc((size_t)(4 * P), sizeof(int32_t));
    int32_t *g2 = (int32_t *)calloc((size_t)(4 * P), sizeof(int32_t));
    float *g3 = (float *)calloc((size_t)(4 * P), sizeof(float));
    float *g4 = (float *)calloc((size_t)(4 * P), sizeof(float));
    for (int64_t i = 0; i < 8 * P; ++i) snt[i] = 0.0;
    for (int64_t p = 0; p < P; ++p) {
        for (int32_t s = S - 1; s >= 0; --s) {
            const int32_t *base = recs + (size_t)s * R * 8;
            if (dup_off)
                for (uint32_t e = dup_off[p + 1]; e > dup_off[p]; --e)
                    visit_record(base + (P + e - 1) * 8, RDC(s, P + e - 1), s, C, cov, p, P, snt, srd2, cnt2, nrec2, g1, g2, g3, g4);
            visit_record(base + p * 8, RDC(s, p), s, C, cov, p, P, snt, srd2, cnt2, nrec2, g1, g2, g3, g4);
        }
    }
    free(srd2); free(cnt2); free(nrec2); free(g1); free(g2); free(g3); free(g4);
#undef RDC
}

/* Ordered combine of two partial tables, L covering earlier samples than R.
 * Sums add (exact inside the envelope); the germ-max state machine of
 * EE:1251-1271 composes as: the first qualifying record overall is L's if L
 * has one, and R's first qualifying record then counts as a "later" record. */
void oracle_acc_merge(int64_t P, double *snt, int64_t *srd, int32_t *cnt, int32_t *nrec,
                      int32_t *gm_n, int32_t *gm_first, float *gm_first_af, float *gm_rest,
                      const double *snt_r, const int64_t *srd_r, const int32_t *cnt_r,
                      const int32_t *nrec_r, const int32_t *gm_n_r, const int32_t *gm_first_r,
                      const float *gm_first_af_r, const float *gm_rest_r)
{
    for (int64_t i = 0; i < 8 * P; ++i) {
        snt[i] += snt_r[i];
        srd[i] += srd_r[i];
    }
    for (int64_t p = 0; p < P; ++p) nrec[p] += nrec_r[p];
    for (int64_t i = 0; i < 4 * P; ++i) {
        cnt[i] += cnt_r[i];
        if (gm_n_r[i] == 0) continue;
        if (gm_n[i] == 0) {
            gm_first[i] = gm_first_r[i];
            gm_first_af[i] = gm_first_af_r[i];
            gm_rest[i] = gm_rest_r[i];
        } else {
            float m = gm_rest[i];
            if (m <= gm_first_af_r[i]) m = gm_first_af_r[i];
            if (m <= gm_rest_r[i]) m = gm_rest_r[i];
            gm_rest[i] = m;
        }
        gm_n[i] += gm_n_r[i];
    }
}

float oracle_text_roundtrip(float r)
{
    char buf[400];
    snprintf(buf, sizeof buf, "%f", (double)r); /* EE:1704 */
    return strtof(buf, NULL);                   /* std::stof, VC:889-890 */
}

void oracle_error_finalize(int64_t P, const double *snt, const int64_t *srd, const int32_t *cnt,
                           const int32_t *nrec, const int32_t *gm_n, const float *gm_rest,
                           float *rate, uint8_t *code, float *thr, double *germ_val,
                           uint8_t *germ_present)
{
    for (int64_t p = 0; p < P; ++p) {
        for (int nt = 0; nt < 4; ++nt) {
            int64_t i = nt * P + p, ifw = (0 * 4 + nt) * P + p, ibw = (1 * 4 + nt) * P + p;
            float r_fw = 0, r_bw = 0;
            uint8_t c;
            /* EE:1659 */
            if (cnt[i] < 0.338 * (size_t)nrec[p]) {
                c = 1;
            } else {
                /* EE:1679-1687 */
                double sum_fw_RD = (double)srd[ifw], sum_bw_RD = (double)srd[ibw];
                double AF_fw = (float)snt[ifw] / (float)sum_fw_RD;
                double AF_bw = (float)snt[ibw] / (float)sum_bw_RD;
                if (isnan(AF_fw) || isnan(AF_bw)) {
                    c = 2;
                } else {
                    c = 0;
                    r_fw = (float)AF_fw;
                    r_bw = (float)AF_bw;
                }
            }
            code[i] = c;
            rate[ifw] = r_fw;
            rate[ibw] = r_bw;
            if (thr) {
                /* table cell "-1_-1" is rewritten to "0.01_0.01" (EE:2680-2684); VC reads
                   the text back with std::stof (VC:889-890) */
                thr[ifw] = c ? strtof("0.01", NULL) : oracle_text_roundtrip(r_fw);
                thr[ibw] = c ? strtof("0.01", NULL) : oracle_text_roundtrip(r_bw);
            }
            /* EE:1251-1271: first qualifying record stores the sentinel (-888 for A,
               EE:1260; 0 for C/G/T, EE:1318/1374/1431); later ones `if(value<=AF) value=AF` */
            if (gm_n[i] == 0) {
                germ_present[i] = 0;
                germ_val[i] = 0;
            } else {
                double v = (nt == 0) ? -888.0 : 0.0;
                if (gm_n[i] > 1 && v <= gm_rest[i]) v = gm_rest[i];
                germ_present[i] = 1;
                germ_val[i] = v;
            }
        }
    }
}

int oracle_format_thr_cell(uint8_t code, float r_fw, float r_bw, int is_ref, char *buf)
{
    if (is_ref) return sprintf(buf, "-2_-2");       /* EE:2670-2673 */
    if (code) return sprintf(buf, "0.01_0.01");     /* EE:2680-2684 */
    return sprintf(buf, "%f_%f", (double)r_fw, (double)r_bw); /* EE:1704 */
}

int oracle_format_germ_cell(uint8_t present, double v, char *buf)
{
    if (!present) return sprintf(buf, "-");         /* EE:2811 */
    return sprintf(buf, "%g", v);                   /* ostream<<double, EE:2815 */
}

/* ------------------------------------------------------------------ */
/* Tumour record evaluation                                            */
/* ------------------------------------------------------------------ */

void oracle_poisson_call(const int32_t *trecs, int64_t P, int64_t E, const uint32_t *ext_pos,
                         int32_t T, const float *thr, const uint8_t *ref_code, int32_t cov,
                         uint8_t *call_mask, double *q, float *af)
{
    oracle_poisson_call_rd(trecs, NULL, P, E, ext_pos, T, thr, ref_code, cov, call_mask, q, af);
}

/* the same with the RD column of every record (VC:752): rdcol [T][P+E], ORACLE_ABSENT where it equals A+C+G+T */
void oracle_poisson_call_rd(const int32_t *trecs, const int32_t *rdcol, int64_t P, int64_t E, const uint32_t *ext_pos,
                            int32_t T, const float *thr, const uint8_t *ref_code, int32_t cov,
                            uint8_t *call_mask, double *q, float *af)
{
    const int64_t R = P + E;
    for (int32_t t = 0; t < T; ++t) {
        for (int64_t r = 0; r < R; ++r) {
            const int32_t *c = trecs + ((size_t)t * R + r) * 8;
            int64_t o = (int64_t)t * R + r;
            call_mask[o] = 0;
            if (q) for (int j = 0; j < 8; ++j) q[o * 8 + j] = -1.0;
            if (af) for (int j = 0; j < 12; ++j) af[o * 12 + j] = 0.0f;
            if (c[0] == ORACLE_ABSENT) continue;
            int64_t p = r < P ? r : (int64_t)ext_pos[r - P];
            /* VC:760-761 */
            int FW = c[0] + c[1] + c[2] + c[3];
            int BW = c[4] + c[5] + c[6] + c[7];
            int RD = (rdcol && rdcol[(size_t)t * R + r] != ORACLE_ABSENT) ? rdcol[(size_t)t * R + r] : FW + BW; /* the RD column, VC:752 */
            int RD_reverse = BW; /* VC:819 */
            if (af) {
                for (int nt = 0; nt < 4; ++nt) {
                    /* VC:772-817 */
                    af[o * 12 + nt * 3 + 0] = (float)(c[nt] + c[4 + nt]) / (float)RD;
                    af[o * 12 + nt * 3 + 1] = FW == 0 ? 0 : (float)c[nt] / (float)FW;
                    af[o * 12 + nt * 3 + 2] = BW == 0 ? 0 : (float)c[4 + nt] / (float)BW;
                }
            }
            int ref = ref_code[p];
            if (ref > 3) continue; /* VC:3290-3293: reference base not A/C/G/T */
            for (int nt = 0; nt < 4; ++nt) {
                if (nt == ref) continue;
                float AF_fw = thr[(0 * 4 + nt) * P + p]; /* VC:887-890 */
                float AF_bw = thr[(1 * 4 + nt) * P + p];
                /* VC:895-896 */
                long double Q_fw = oracle_score(c[nt], RD - RD_reverse, AF_fw);
                long double Q_bw = oracle_score(c[4 + nt], RD_reverse, AF_bw);
                if (q) {
                    q[o * 8 + nt * 2 + 0] = (double)Q_fw;
                    q[o * 8 + nt * 2 + 1] = (double)Q_bw;
                }
                /* VC:898 */
                if (FW >= cov && BW >= cov && Q_fw >= 5 && Q_bw >= 5) call_mask[o] |= (uint8_t)(1u << nt);
            }
        }
    }
}
