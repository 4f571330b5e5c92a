// amplisolve_amd/csrc/ampli_math.h
//
// Scalar arithmetic of the hot path, shared by the HIP kernels (device) and by
// the C++ host (formatting / CPU-side unit checks of the same code).  Each
// function names the reference lines whose result it must reproduce:
//   EE:n = source_codes/AmpliSolveErrorEstimation.cpp:n, VC:n = source_codes/AmpliSolveVariantCalling.cpp:n
#ifndef AMPLI_MATH_H
#define AMPLI_MATH_H
#include <math.h>
#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define AMPLI_FN __host__ __device__ __forceinline__
#else
#define AMPLI_FN static inline
#endif

// ---------------------------------------------------------------------------
// AF gate.  The reference computes af = float(x)/float(d), widens it to double
// and tests af <= 0.05 (EE:1229,1251 and EE:1592-1595).  0.05 (double) lies
// strictly between the adjacent floats f0 = 13421772*2^-28 and f1 = f0 + 2^-28,
// so the test is RN(x/d) <= f0.  With round-to-nearest-even and f0's even
// significand that is x/d <= (f0+f1)/2 = 26843545 * 2^-29, i.e. for exactly
// representable x, d (< 2^24):   x <= floor(d * 26843545 / 2^29).
// One 64-bit multiply per (record, strand) instead of one IEEE division per
// (record, strand, nucleotide).  d == 0 gives limit 0; callers AND the gate
// with d >= coverage_cutoff >= 1 as the reference does, which also removes
// the 0/0 = NaN case.
// ---------------------------------------------------------------------------
#define AMPLI_AF_MID_NUM 26843545ull
#define AMPLI_AF_MID_SHIFT 29
#define AMPLI_COUNT_LIMIT (1 << 24)

// floor(d * 26843545 / 2^29) = floor(d * (26843545 * 8) / 2^32): the high half of ONE 32 x 32-bit multiply (v_mul_hi_u32)
// instead of a 64-bit multiply-add and a 64-bit shift; 26843545 * 8 < 2^32, so the identity holds for every uint32 d.
#define AMPLI_AF_MID_NUM8 214748360u
AMPLI_FN int32_t ampli_af_limit(int32_t d)
{
    return (int32_t)(((uint64_t)(uint32_t)d * AMPLI_AF_MID_NUM8) >> 32);
}

// The same integer as one fp32 multiply and a truncation, for 0 <= d < 2^24 (d exact in fp32): 0x1.999998p-5 is the float just
// below 0.05, and the rounding error of the product never carries it across an integer that the exact quotient d * 26843545 / 2^29
// has not crossed (tests/test_math_host.py compares every d).  Full rate on the GPU where v_mul_hi_u32 takes four passes; the
// float of d is needed by the callers anyway.  Negative d (an absent record's sentinel sum) gives a value <= 0: callers mask it.
AMPLI_FN int32_t ampli_af_limit_f32(float d_as_float)
{
    return (int32_t)(d_as_float * 0x1.999998p-5f);
}

// literal form, for operands outside the exact-float range and for tests
AMPLI_FN int ampli_af_gate_fp(int32_t x, int32_t d)
{
    double af = (double)((float)x / (float)d);
    return af <= 0.05;
}

// ---------------------------------------------------------------------------
// stof(sprintf("%f", r)): the rate as AmpliSolveVariantCalling reads it back
// from the error table (EE:1704 -> VC:887-890), in exact integer arithmetic.
//   "%f" prints round-half-even(r * 10^6) / 10^6 (glibc rounds the exact binary
//   value); r = m * 2^e with m < 2^24, so m * 10^6 < 2^44 fits an int64.
//   For r >= 16 the float spacing (>= 2^-19) exceeds 2 * 5e-7, so the nearest
//   float to the printed decimal is r itself.  Below 16, N < 2^24 and 10^6 are
//   both exact floats and the correctly rounded fp32 quotient N / 10^6 is the
//   float nearest to the printed decimal, which is what strtof returns.
// ---------------------------------------------------------------------------
AMPLI_FN float ampli_text_roundtrip(float r)
{
    uint32_t bits;
    memcpy(&bits, &r, 4);
    const uint32_t sign = bits & 0x80000000u;
    const uint32_t ex = (bits >> 23) & 0xFF;
    uint32_t man = bits & 0x7FFFFFu;
    if (ex == 0xFF) return r; // inf / nan: not produced by the reference's finalize (NaN is caught before)
    float a = fabsf(r);
    if (a >= 16.0f) return r;
    int e2;
    if (ex == 0) {
        e2 = -149;
    } else {
        man |= 0x800000u;
        e2 = (int)ex - 150;
    }
    // a = man * 2^e2, e2 <= -20 here
    const uint64_t A = (uint64_t)man * 1000000ull;
    const int k = -e2;
    uint64_t N;
    if (k >= 45) {
        N = 0; // A < 2^44 <= 2^(k-1): rounds to 0
    } else {
        const uint64_t q = A >> k, rem = A & ((1ull << k) - 1), half = 1ull << (k - 1);
        N = q + ((rem > half) || (rem == half && (q & 1)));
    }
    float out = (float)(uint32_t)N / 1000000.0f;
    uint32_t ob;
    memcpy(&ob, &out, 4);
    ob |= sign; // "-0.000000" parses to -0.0f
    memcpy(&out, &ob, 4);
    return out;
}

// ---------------------------------------------------------------------------
// Poisson scorer: kf_lgamma / _kf_gammap / _kf_gammaq / kf_gammaq as adopted by
// the reference from samtools kfunc.c (VC:3721-3830), including the 99-step
// caps, KF_GAMMA_EPS 1e-14 and KF_TINY 1e-290 (VC:149-150).
// ---------------------------------------------------------------------------
AMPLI_FN double ampli_kf_lgamma(double z)
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

AMPLI_FN double ampli_kf_gammap_series(double s, double z)
{
    double sum = 1., x = 1.;
    for (int k = 1; k < 100; ++k) {
        x *= z / (s + k);
        sum += x;
        if (x / sum < 1e-14) break;
    }
    return exp(s * log(z) - z - ampli_kf_lgamma(s + 1.) + log(sum));
}

AMPLI_FN double ampli_kf_gammaq_cf(double s, double z)
{
    double f = 1. + z - s, C = f, D = 0.;
    for (int j = 1; j < 100; ++j) {
        const double a = j * (s - j), b = (j << 1) + 1 + z - s;
        D = b + a * D;
        if (D < 1e-290) D = 1e-290;
        C = b + a / C;
        if (C < 1e-290) C = 1e-290;
        D = 1. / D;
        const double d = C * D;
        f *= d;
        if (fabs(d - 1.) < 1e-14) break;
    }
    return exp(s * log(z) - z - ampli_kf_lgamma(s) - log(f));
}

// The same series without its 99 divisions, for the dense drain of the prefilter queue (z < s there):
//   sum_{n=0}^{99} z^n / ((s+1)...(s+n)) = A_99 / D_99,   A_n = A_{n-1} (s+n) + z^n,   D_n = D_{n-1} (s+n),
// one fused multiply-add per term on the critical path instead of an IEEE division (~10 dependent operations), every
// term positive (no cancellation), A / D in [1, 100]; A, D and z^n are rescaled by an exact power of two every 16 terms
// ((s+n)^16 < 2^500 for every int32 count).  Differences from ampli_kf_gammap_series: (i) rounding, ~1e-14 relative;
// (ii) the reference stops at the first term with x/sum < 1e-14 -- the terms it leaves out add < ~1e-13 of the sum
// (this form stops, between runs of 16 terms, only once a term is below 2^-60 of the sum).
// Both are eight orders of magnitude inside the 1e-6 tolerance on p.  What is NOT optional is the cap at n = 99:
// for z close to s the series has not converged by then and the truncation is part of the reference's result.
AMPLI_FN double ampli_kf_gammap_series_nodiv(double s, double z)
{
    double A = 1., D = 1., zp = 1., t = s;
    // terms 1..99 as six runs of 16 and one of 3; the rescale sits BETWEEN the runs so that a run is four
    // operations per term and nothing else
    for (int blk = 0; blk < 7; ++blk) {
        const int len = blk < 6 ? 16 : 3;
        for (int i = 0; i < len; ++i) {
            t += 1.;
            zp *= z;
            A = fma(A, t, zp);
            D *= t;
        }
        // z < s: every further term is smaller than the last one (zp / D), so once that is below 2^-60 of the sum the
        // at most 83 terms still to come cannot change the double any more
        if (zp < A * 8.673617379884035e-19) break;
        int e;
        (void)frexp(D, &e);
        A = ldexp(A, -e); D = ldexp(D, -e); zp = ldexp(zp, -e);
    }
    return exp(s * log(z) - z - ampli_kf_lgamma(s + 1.) + log(A / D)); // VC:3793
}

AMPLI_FN double ampli_kf_gammaq(double s, double z)
{
    return (z <= 1. || z < s) ? 1. - ampli_kf_gammap_series(s, z) : ampli_kf_gammaq_cf(s, z);
}

// p as formed at VC:3858-3865 (before the clamp); err already != -1
AMPLI_FN double ampli_poisson_p(int32_t k, int32_t rd, float err)
{
    if (err == 0) err = 0.0010008f; // VC:3852-3856 (double literal narrowed to float)
    if (k == 0) return 1.0;         // VC:3858-3861
    const double m = (double)rd * err; // VC:3864: double * float
    return 1 - ampli_kf_gammaq((double)k, m);
}

// Q = -10 log10 p with the reference's clamps (VC:3844-3880).  The reference
// takes the final log10 in long double; fp64 differs by < 1e-14 relative.
AMPLI_FN double ampli_q_from_p(double p)
{
    if (p < 0.0000000001) return 100.0; // -10*log10l(1e-10), (double) of which is 100
    if (p == 1) return 0.0;
    return -10 * log10(p);
}

AMPLI_FN double ampli_poisson_score(int32_t k, int32_t rd, float err)
{
    if (err == -1) return -888.0; // VC:3844-3849
    return ampli_q_from_p(ampli_poisson_p(k, rd, err));
}

// ---------------------------------------------------------------------------
// The all-scores mode's scorer (round 4): the same Q as ampli_poisson_score, to rounding, for what that mode feeds it --
// s = k is always an INTEGER count -- at about a third of the instructions:
//   * kf_lgamma(s), kf_lgamma(s + 1) come from a table of the Lanczos form's own values at the integers (lgtab[n] =
//     ampli_kf_lgamma(n), filled by the same function on the same device, so bit-identical to calling it; NULL or an index
//     beyond ntab: the function itself): 8 divisions + 2 logarithms less per score;
//   * the series branch is the division-free form of the drain kernel (ampli_kf_gammap_series_nodiv);
//   * the continued fraction (VC:3733-3752) is evaluated through its convergents, f_j = A_j / B_j with
//       A_j = b_j A_{j-1} + a_j A_{j-2},  B_j = b_j B_{j-1} + a_j B_{j-2},  a_j = j (s - j),  b_j = 2j + 1 + z - s
//     (modified Lentz keeps C_j = A_j / A_{j-1} and D_j = B_{j-1} / B_j and pays two divisions per step; the product of the
//     C_j D_j it accumulates IS A_j / B_j).  The reference leaves the loop once |C_j D_j - 1| < 1e-14 or after 99 steps, and
//     for an integer s at j = s at the latest (a_s = 0 makes the step the identity); all three exits are kept.  In this
//     branch z > s >= 1, so every a_j up to the exit and every b_j is positive and the KF_TINY guards never fire;
//   * log(sum) / log(f) are not taken: the prefactor exp(s log z - z - lgamma) is multiplied / divided instead.
// Differences from the literal scorer: rounding (~1e-13 relative on p, measured in tests/test_math_host.py); the contract
// is 1e-6, and a Q within 1e-6 of a gate is re-decided by the host in the reference's own arithmetic either way.
// ---------------------------------------------------------------------------
AMPLI_FN double ampli_lgamma_int(int n, const double *lgtab, int ntab)
{
    return (lgtab && n < ntab) ? lgtab[n] : ampli_kf_lgamma((double)n);
}

// P(s, z) by the series, z <= s (the branch kf_gammaq takes for z <= 1 or z < s), prefactor given
AMPLI_FN double ampli_gammap_series_int(double s, double z, double prefactor /* exp(s log z - z - lgamma(s + 1)) */)
{
    double A = 1., D = 1., zp = 1., t = s;
    for (int blk = 0; blk < 7; ++blk) {
        const int len = blk < 6 ? 16 : 3;
        for (int i = 0; i < len; ++i) {
            t += 1.;
            zp *= z;
            A = fma(A, t, zp);
            D *= t;
        }
        if (zp < A * 8.673617379884035e-19) break; // see ampli_kf_gammap_series_nodiv
        int e;
        (void)frexp(D, &e);
        A = ldexp(A, -e); D = ldexp(D, -e); zp = ldexp(zp, -e);
    }
    return prefactor * (A / D);
}

// Q(s, z) by the continued fraction for an integer s = k, z > s, prefactor given
AMPLI_FN double ampli_gammaq_cf_int(int k, double z, double prefactor /* exp(s log z - z - lgamma(s)) */)
{
    const double s = (double)k;
    double A1 = 1. + z - s, A2 = 1., B1 = 1., B2 = 0.; // f_0 = b_0; A_{-1} = 1, B_{-1} = 0, B_0 = 1
    const int J = k - 1 < 99 ? k - 1 : 99;             // j = s is the identity step; the reference's cap is j < 100
    for (int j = 1; j <= J; ++j) {
        const double a = (double)j * (s - j), b = (double)((j << 1) + 1) + z - s;
        const double A0 = fma(b, A1, a * A2), B0 = fma(b, B1, a * B2);
        const double num = A0 * B1, den = A1 * B0; // C_j D_j = num / den, both positive
        A2 = A1; A1 = A0; B2 = B1; B1 = B0;
        if (fabs(num - den) < 1e-14 * den) break;
        if ((j & 7) == 0) { // a, b < 2^33: eight steps stay far inside the double range
            int e;
            (void)frexp(A1, &e);
            A1 = ldexp(A1, -e); A2 = ldexp(A2, -e); B1 = ldexp(B1, -e); B2 = ldexp(B2, -e);
        }
    }
    return prefactor * B1 / A1; // exp(...) / f, f = A_j / B_j  (VC:3751)
}

// Q(k, z) for an integer k in 1 .. AMPLI_HORNER_K on the continued fraction's side (z > k), round 5.  For an integer s the
// fraction of VC:3733-3752 ends at j = s (a_s = 0: the step is the identity) and its value there is a rational function,
//   f = z^k / H_k(z),   H_k(z) = sum_{i<k} z^i (k-1)! / i!   (H_1 = 1, H_{j+1} = j H_j + z^j),
// so exp(s log z - z - lgamma(s)) / f  =  exp(-z - lgamma(k)) H_k(z): the Poisson sum e^-z sum_{i<k} z^i / i! with the
// reference's own lgamma (the table of its Lanczos values).  One exp and k - 1 FMAs -- no log z, no division, no convergence
// test; the reference's early exit (|C_j D_j - 1| < 1e-14) leaves out steps that change f by < 1e-14 each.  Within 1e-13
// relative of the literal scorer (tests/test_math_host.py).  K = 4 covers 96 % of the scores on that side of a ctDNA-like panel
// (config 3: all-scores mode 0.55 ms; K = 16: 0.58 ms -- a wave runs the longest loop of its lanes; H_16 would still be far
// from overflow for z < 2^24 x 0.05, the bound kept below).
#define AMPLI_HORNER_K 4
AMPLI_FN double ampli_gammaq_horner_int(int k, double z, double lgk /* lgamma(k) */)
{
    double H = 1., zp = 1.;
    for (int j = 1; j < k; ++j) {
        zp *= z;
        H = fma((double)j, H, zp);
    }
    return exp(-z - lgk) * H;
}

// p of a count k > 0 against an effective error (neither -1 nor 0): VC:3864-3865 through the forms above
AMPLI_FN double ampli_poisson_p_dense(int32_t k, int32_t rd, float err, const double *lgtab, int ntab)
{
    if (k < 0) return 1 - ampli_kf_gammaq((double)k, (double)rd * err); // not a count: the literal path
    const double s = (double)k, z = (double)rd * err; // VC:3864: double * float
    if (z <= 1. || z < s) { // VC:3728
        if (!(z > 0)) return 1 - ampli_kf_gammaq(s, z); // z <= 0 or NaN: whatever the literal arithmetic gives
        const double P = ampli_gammap_series_int(s, z, exp(s * log(z) - z - ampli_lgamma_int(k + 1, lgtab, ntab)));
        return 1 - (1. - P); // VC:3865 on top of VC:3728
    }
    if (k <= AMPLI_HORNER_K && z < 838860.8) return 1 - ampli_gammaq_horner_int(k, z, ampli_lgamma_int(k, lgtab, ntab));
    return 1 - ampli_gammaq_cf_int(k, z, exp(s * log(z) - z - ampli_lgamma_int(k, lgtab, ntab)));
}

// The drain kernel's scorer since round 6: p of a queued item -- an integer count k against a mean 0 < m < k, i.e. always the
// series branch of kf_gammaq (VC:3728) -- in the all-scores mode's form: kf_lgamma(k + 1) from the table of its own values, the
// prefactor multiplied by the series' sum instead of adding its logarithm.  A wave of the drain is bound by the length of ONE
// instruction stream (a pass is ~1500 dependent fp64 instructions at ~8 cycles): this form has a third fewer.
AMPLI_FN double ampli_drain_p(int32_t k, double m, const double *lgtab, int ntab)
{
    const double s = (double)k;
    const double P = ampli_gammap_series_int(s, m, exp(s * log(m) - m - ampli_lgamma_int(k + 1, lgtab, ntab)));
    return 1 - (1. - P); // VC:3865 on top of VC:3728
}

AMPLI_FN double ampli_poisson_score_dense(int32_t k, int32_t rd, float err, const double *lgtab, int ntab)
{
    if (err == -1) return -888.0;   // VC:3844-3849
    if (err == 0) err = 0.0010008f; // VC:3852-3856
    if (k == 0) return 0.0;         // VC:3858-3861: p = 1 -> Q = 0 (VC:3873-3876)
    return ampli_q_from_p(ampli_poisson_p_dense(k, rd, err, lgtab, ntab));
}

// Exact-decision bound used by AMPLI_POISSON_PREFILTER: when k <= m the
// reference's own scorer returns Q < 5 (P(X >= k) > 0.31 for k <= mean; checked
// exhaustively against the scorer incl. its iteration caps in
// tests/test_math_host.py), so VC:898 is false whatever the other strand says.
// err == -1 gives Q = -888 < 5 as well.
AMPLI_FN int ampli_prefilter_nocall(int32_t k, int32_t rd, float err)
{
    if (err == -1) return 1;
    if (err == 0) err = 0.0010008f;
    if (k == 0) return 1;
    const double m = (double)rd * err;
    return (double)k <= m;
}

// Conservative fp32 form of the bound above, used by the streaming kernel: with err_eff = ampli_effective_err(err)
// and c = float(rd) * 0.999999f,   float(k) <= c * err_eff   implies   k <= rd * err   (k, rd < 2^24 are exact
// floats; two roundings of at most 2^-24 each cannot undo the factor 1 - 1e-6), so every record it skips would be
// skipped by ampli_prefilter_nocall too.  What it does not skip is scored exactly; nothing is decided in fp32.
AMPLI_FN float ampli_effective_err(float err)
{
    if (err == 0.0f) return 0.0010008f; // VC:3852-3856
    if (err == -1.0f) return INFINITY;  // VC:3844-3849: Q = -888
    return err;
}

AMPLI_FN int ampli_prefilter_skip_f32(int32_t k, int32_t rd, float err_eff)
{
    if ((uint32_t)rd >= (uint32_t)AMPLI_COUNT_LIMIT || (uint32_t)k >= (uint32_t)AMPLI_COUNT_LIMIT) return 0;
    const float c = (float)rd * 0.999999f;
    return (float)k <= c * err_eff;
}

#endif
