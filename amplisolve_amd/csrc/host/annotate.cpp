// amplisolve_amd/csrc/host/annotate.cpp -- post-call annotation of the (sparse) emitted calls:
// Fisher strand-bias test, +-10-mer context, homopolymer flag (VC:3307-3718, VC:3797-3814).
#include <algorithm>
#include <cmath>

#include "../ampli_math.h"
#include "host.hpp"

namespace ampli {

// Decision guard for calls within rounding of the gates (include/amplisolve_hip.h, AMPLI_CALL_BORDERLINE): the reference's
// own operation sequence for Q -- kf_gammaq in double with the host's libm, the final log10 in x87 long double
// (VC:3834-3884, line for line: err == -1 -> -888, err == 0 -> 0.0010008f, k == 0 -> p = 1, p < 1e-10 -> -10*log10l(1e-10),
// p == 1 -> 0).  Only ever called for the handful of emitted pairs whose device Q lies within 1e-6 of 5 or of 20.
long double score_reference_sequence(int k, int rd, float err)
{
    if (err == -1) return -888;
    if (err == 0) err = 0.0010008;
    double p;
    if (k == 0) p = 1;
    else {
        const double m = double(rd) * err;
        p = 1 - ampli_kf_gammaq(k, m);
    }
    long double Q;
    if (p < 0.0000000001) {
        p = 0.0000000001;
        Q = -10 * log10l(p);
    } else if (p == 1) {
        Q = 0;
    } else {
        Q = -10 * log10l(p);
    }
    return Q;
}

// Two-sided Fisher exact test as VC:3797-3814 forms it: N = a+b+c+d, r = a+c, n = c+d, sum of the
// hypergeometric pmf over all k whose probability does not exceed that of the observed k = c.
// The reference takes the pmf from Boost.Math (absent here, not vendored by the reference): this is an own
// log-gamma pmf; probabilities within 1e-7 relative of the observed one count as ties (as R's fisher.test).
// PARITY UNPINNED at the Boost boundary.
double fisher_two_sided(int a, int b, int c, int d)
{
    const unsigned N = (unsigned)(a + b + c + d), r = (unsigned)(a + c), n = (unsigned)(c + d);
    const unsigned max_for_k = std::min(r, n);
    const unsigned min_for_k = (unsigned)std::max(0, (int)(r + n - N));
    auto lchoose = [](double nn, double kk) { return std::lgamma(nn + 1) - std::lgamma(kk + 1) - std::lgamma(nn - kk + 1); };
    const double ldenom = lchoose(N, n);
    auto pmf = [&](unsigned k) { return std::exp(lchoose(r, k) + lchoose((double)N - r, (double)n - k) - ldenom); };
    const double cutoff = pmf((unsigned)c) * (1 + 1e-7);
    if (max_for_k < min_for_k) return 0.0;
    // The support can be thousands of k wide (r, n of the order of the read depth) and only the pmf values are needed, so
    // the log-gamma form is evaluated once, at the mode, and the rest follows from the ratio of neighbouring terms
    //   pmf(k+1) / pmf(k) = (r-k)(n-k) / ((k+1)(N-r-n+k+1))
    // walked outwards in both directions (each step one multiplication and one division; ~1e-13 relative after a few
    // thousand steps, six orders inside the 1e-7 tie tolerance and the six digits that are printed).
    const double Nd = N, rd = r, nd = n;
    unsigned mode = (unsigned)std::floor((rd + 1) * (nd + 1) / (Nd + 2));
    mode = std::min(std::max(mode, min_for_k), max_for_k);
    const double p_mode = pmf(mode);
    double tmp_p = p_mode <= cutoff ? p_mode : 0.0;
    double p = p_mode;
    for (unsigned k = mode; k < max_for_k; ++k) { // upwards: pmf(k+1) from pmf(k)
        p *= ((rd - k) * (nd - k)) / (((double)k + 1) * (Nd - rd - nd + k + 1));
        if (p <= cutoff) tmp_p += p;
        if (p == 0) break;
    }
    p = p_mode;
    for (unsigned k = mode; k > min_for_k; --k) { // downwards: pmf(k-1) from pmf(k)
        p *= ((double)k * (Nd - rd - nd + k)) / ((rd - k + 1) * (nd - k + 1));
        if (p <= cutoff) tmp_p += p;
        if (p == 0) break;
    }
    return tmp_p;
}

// the same sum with every term from the log-gamma form (round 1-3; kept as the check of the recurrence: tests/test_host_logic.py)
double fisher_two_sided_direct(int a, int b, int c, int d)
{
    const unsigned N = (unsigned)(a + b + c + d), r = (unsigned)(a + c), n = (unsigned)(c + d);
    const unsigned max_for_k = std::min(r, n);
    const unsigned min_for_k = (unsigned)std::max(0, (int)(r + n - N));
    auto lchoose = [](double nn, double kk) { return std::lgamma(nn + 1) - std::lgamma(kk + 1) - std::lgamma(nn - kk + 1); };
    const double ldenom = lchoose(N, n);
    auto pmf = [&](unsigned k) { return std::exp(lchoose(r, k) + lchoose((double)N - r, (double)n - k) - ldenom); };
    const double cutoff = pmf((unsigned)c);
    double tmp_p = 0.0;
    for (unsigned k = min_for_k; k < max_for_k + 1; ++k) {
        const double p = pmf(k);
        if (p <= cutoff * (1 + 1e-7)) tmp_p += p;
    }
    return tmp_p;
}

static std::string base_or(const Panel &p, const std::string &chrom, int pos, const char *missing)
{
    const int i = p.find(chrom, pos);
    return i < 0 ? std::string(missing) : p.ref_base[i];
}

// offsets -10..-1; a position outside the panel contributes "-|", except offsets -6, -3 and -1 which
// contribute a bare "-" (VC:3389, 3425, 3449)
std::string kmer_down(const Panel &p, const std::string &chrom, int pos)
{
    std::string s;
    for (int o = 10; o >= 1; --o) s += base_or(p, chrom, pos - o, (o == 6 || o == 3 || o == 1) ? "-" : "-|");
    return s;
}

// offsets +1..+10; missing -> "-|", except +10 -> "-" (VC:3604)
std::string kmer_up(const Panel &p, const std::string &chrom, int pos)
{
    std::string s;
    for (int o = 1; o <= 10; ++o) s += base_or(p, chrom, pos + o, o == 10 ? "-" : "-|");
    return s;
}

// VC:3615-3718: count A/C/G/T over both context strings plus the substituted base; flag when the two most
// frequent bases of any pair exceed 18 of the 21
int homopolymer_test(const std::string &down, const std::string &up, char sub)
{
    int n[4] = {0, 0, 0, 0};
    auto add = [&](char ch) {
        if (ch == 'A') ++n[0];
        else if (ch == 'C') ++n[1];
        else if (ch == 'G') ++n[2];
        else if (ch == 'T') ++n[3];
    };
    add(sub);
    for (char ch : down) add(ch);
    for (char ch : up) add(ch);
    if (n[0] + n[1] > 18 || n[0] + n[2] > 18 || n[0] + n[3] > 18 || n[1] + n[2] > 18 || n[1] + n[3] > 18 || n[3] + n[2] > 18) return 1;
    return 0;
}

} // namespace ampli
