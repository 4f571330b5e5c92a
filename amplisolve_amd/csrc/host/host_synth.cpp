// amplisolve_amd/csrc/host/host_synth.cpp -- host instantiation of the synthetic panel generator and
// of the scalar helpers shared with the kernels (csrc/ampli_math.h).
#include <sys/stat.h>

#include <atomic>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

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
        const double pv = ampli_drain_p(k[i], m, nullptr, 0); // no table on the host: the same values from the function itself
        if (p) p[i] = pv;
        if (q) q[i] = ampli_q_from_p(pv);
    }
}
extern "C" void ampli_host_dense_score_batch(const int32_t *k, const int32_t *rd, const float *err, int64_t n, double *q)
{
    for (int64_t i = 0; i < n; ++i) q[i] = ampli_poisson_score_dense(k[i], rd[i], err[i], nullptr, 0);
}
extern "C" void ampli_host_af_limit_batch(const int32_t *d, int64_t n, int32_t *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = ampli_af_limit(d[i]);
}

extern "C" void ampli_host_af_limit_f32_batch(const int32_t *d, int64_t n, int32_t *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = ampli_af_limit_f32((float)d[i]);
}

// ---------------------------------------------------------------------------------------------------------
// The synthetic panel as FILES (bench.py's end-to-end leg and the file-level tests): BED, reference-base table and one
// .PILEUP.ASEQ per sample with exactly the counts ampli_synth_fill puts into HBM.  Panel = 30 amplicon regions on
// chr1..chr22, chrX cycling, ceil(P/30) consecutive 1-based positions each.
// ---------------------------------------------------------------------------------------------------------
namespace {

struct Region { std::string chrom; int64_t start, n; };

std::vector<Region> synth_regions(int64_t P)
{
    std::vector<Region> out;
    const int64_t per = (P + 29) / 30;
    for (int i = 0; i < 30; ++i) {
        const int64_t n = std::min<int64_t>(per, P - (int64_t)i * per);
        if (n <= 0) break;
        const int c = i % 23;
        out.push_back(Region{c == 22 ? std::string("chrX") : "chr" + std::to_string(c + 1), 1000000 + (int64_t)(i / 23) * 5000000, n});
    }
    return out;
}

inline char *put_u(char *p, uint64_t v)
{
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    return p;
}

} // namespace

extern "C" int ampli_host_synth_write_panel(const char *bed_path, const char *refbases_path, int64_t P, uint64_t seed)
{
    if (P <= 0) return -1;
    const auto regs = synth_regions(P);
    if (bed_path) {
        FILE *f = fopen(bed_path, "w");
        if (!f) return -1;
        int i = 0;
        for (auto &r : regs) { fprintf(f, "%s\t%lld\t%lld\tAMPL%d\trs%d\tGENE%d\n", r.chrom.c_str(), (long long)r.start, (long long)(r.start + r.n - 1), i, i, i); ++i; }
        fclose(f);
    }
    if (refbases_path) {
        FILE *f = fopen(refbases_path, "w");
        if (!f) return -1;
        int64_t p = 0;
        for (auto &r : regs)
            for (int64_t j = 0; j < r.n; ++j, ++p) fprintf(f, "%s\t%lld\t%c\n", r.chrom.c_str(), (long long)(r.start + j), "ACGT"[ampli_synth_ref_base(seed, (uint64_t)p)]);
        fclose(f);
    }
    return 0;
}

extern "C" int64_t ampli_host_synth_write_aseq(const char *dir, const char *prefix, int64_t P, int32_t n_samples, int32_t first_sample,
                                               uint64_t seed, int32_t depth, int32_t tumour, int32_t n_threads)
{
    if (!dir || !prefix || P <= 0 || n_samples <= 0 || depth <= 0) return -1;
    mkdir(dir, 0777);
    const auto regs = synth_regions(P);
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    n_threads = std::max(1, std::min(n_threads, (int)n_samples));
    std::atomic<int> next{0};
    std::atomic<int64_t> bytes{0};
    std::atomic<int> failed{0};
    auto work = [&] {
        std::vector<char> buf;
        for (int s; (s = next.fetch_add(1)) < n_samples;) {
            buf.clear();
            buf.reserve((size_t)P * 56 + 128);
            const char *hdr = "chr\tpos\tdbsnp\tMAF\tref\talt\tA\tC\tG\tT\tRD\tArs\tCrs\tGrs\tTrs\n";
            buf.insert(buf.end(), hdr, hdr + strlen(hdr));
            int64_t p = 0;
            char line[256];
            for (auto &r : regs)
                for (int64_t j = 0; j < r.n; ++j, ++p) {
                    int32_t rec[8];
                    ampli_synth_record(seed, (uint64_t)p, (uint64_t)(first_sample + s), depth, tumour, rec);
                    if (rec[0] == AMPLI_SYNTH_ABSENT) continue;
                    char *q = line;
                    memcpy(q, r.chrom.data(), r.chrom.size()); q += r.chrom.size();
                    *q++ = '\t'; q = put_u(q, (uint64_t)(r.start + j));
                    memcpy(q, "\t.\t.\t.\t.", 8); q += 8;
                    int64_t rd = 0;
                    for (int nt = 0; nt < 4; ++nt) { *q++ = '\t'; q = put_u(q, (uint64_t)(rec[nt] + rec[4 + nt])); rd += rec[nt] + rec[4 + nt]; }
                    *q++ = '\t'; q = put_u(q, (uint64_t)rd);
                    for (int nt = 0; nt < 4; ++nt) { *q++ = '\t'; q = put_u(q, (uint64_t)rec[4 + nt]); }
                    *q++ = '\n';
                    buf.insert(buf.end(), line, q);
                }
            char name[64];
            snprintf(name, sizeof name, "/%s%05d.PILEUP.ASEQ", prefix, first_sample + s);
            FILE *f = fopen((std::string(dir) + name).c_str(), "w");
            if (!f || fwrite(buf.data(), 1, buf.size(), f) != buf.size()) failed = 1;
            if (f) fclose(f);
            bytes += (int64_t)buf.size();
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    return failed ? -1 : bytes.load();
}
