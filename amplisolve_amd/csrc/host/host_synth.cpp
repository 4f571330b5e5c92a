// amplisolve_amd/csrc/host/host_synth.cpp -- host instantiation of the synthetic panel generator and
// of the scalar helpers shared with the kernels (csrc/ampli_math.h).
#include "../../../include/amplisolve_host.h"
#include "../ampli_math.h"
#include "../ampli_synth.h"

extern "C" int ampli_host_synth_fill(int32_t *recs, int64_t P, int32_t n_samples, int32_t first_sample, uint64_t seed,
                                     int32_t depth, int32_t tumour)
{
    if (!recs || P <= 0 || n_samples <= 0 || depth <= 0) return -1;
    for (int32_t s = 0; s < n_samples; ++s)
        for (int64_t p = 0; p < P; ++p)
            ampli_synth_record(seed, (uint64_t)p, (uint64_t)(first_sample + s), depth, tumour, recs + ((size_t)s * P + p) * 8);
    return 0;
}

extern "C" int ampli_host_synth_ref(uint8_t *ref_code, int64_t P, uint64_t seed)
{
    if (!ref_code || P <= 0) return -1;
    for (int64_t p = 0; p < P; ++p) ref_code[p] = (uint8_t)ampli_synth_ref_base(seed, (uint64_t)p);
    return 0;
}

extern "C" void ampli_host_text_roundtrip_batch(const float *in, int64_t n, float *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = ampli_text_roundtrip(in[i]);
}

extern "C" int32_t ampli_host_af_limit(int32_t d) { return ampli_af_limit(d); }
extern "C" int ampli_host_prefilter_nocall(int32_t k, int32_t rd, float err) { return ampli_prefilter_nocall(k, rd, err); }
extern "C" int ampli_host_prefilter_skip_f32(int32_t k, int32_t rd, float err) { return ampli_prefilter_skip_f32(k, rd, ampli_effective_err(err)); }
// the drain kernel's scorer for one strand of a queued item (k > m = rd*err > 0): Q of p = 1 - (1 - P_series_nodiv(k, m))
extern "C" void ampli_host_drain_score_batch(const int32_t *k, const int32_t *rd, const float *err, int64_t n, double *q, double *p)
{
    for (int64_t i = 0; i < n; ++i) {
        const double m = (double)rd[i] * err[i];
        const double pv = 1 - (1. - ampli_kf_gammap_series_nodiv((double)k[i], m));
        if (p) p[i] = pv;
        if (q) q[i] = ampli_q_from_p(pv);
    }
}
extern "C" void ampli_host_af_limit_batch(const int32_t *d, int64_t n, int32_t *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = ampli_af_limit(d[i]);
}
