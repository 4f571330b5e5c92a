// oracle/ref_vc_scorer_export.cpp -- TEST INFRASTRUCTURE (reference build glue), not product code.
//
// C-linkage exports over the reference's own Poisson scorer functions, which
// oracle/Makefile compiles from AmpliSolveVariantCalling.cpp WHERE IT LIES
// (line ranges VC:136-137,149-150,164-170,3720-3795,3816-3884 piped to g++;
// nothing but the object file is written).  The complete translation unit
// cannot be built here: VC:135 includes Boost.Math, which this image lacks,
// and no stand-in header is written or used.  The scorer itself is Boost-free.
double kf_lgamma(double z);                                                       // VC:3817
double kf_gammaq(double s, double z);                                             // VC:3726
long double mutationRulesPoissonQualityScore(int supporting_reads, int RD, float AF_error); // VC:3834

extern "C" {
double ref_kf_lgamma(double z) { return kf_lgamma(z); }
double ref_kf_gammaq(double s, double z) { return kf_gammaq(s, z); }
double ref_score(int k, int rd, float err) { return (double)mutationRulesPoissonQualityScore(k, rd, err); }
void ref_score_ld(int k, int rd, float err, long double *out) { *out = mutationRulesPoissonQualityScore(k, rd, err); }
void ref_score_batch(const int *k, const int *rd, const float *err, long n, double *q)
{
    for (long i = 0; i < n; ++i) q[i] = (double)mutationRulesPoissonQualityScore(k[i], rd[i], err[i]);
}
void ref_gammaq_batch(const double *s, const double *z, long n, double *out)
{
    for (long i = 0; i < n; ++i) out[i] = kf_gammaq(s[i], z[i]);
}
}
