// amplisolve_amd/csrc/ampli_synth.h
//
// Deterministic synthetic amplicon panel: counter-based,
// integer-only, so the ASEQ text writer, the host SoA builder and the on-device
// generator (ampli_synth_fill) produce bit-identical counts.  Nothing here comes
// from the reference; it only produces inputs in the shape the reference reads
// (.PILEUP.ASEQ columns, EE:1149).
//
// Per (position p, sample s): total depth = nominal depth x per-sample scale x
// per-position efficiency x jitter; forward share 45-55 %; per (p, alt nt, strand)
// an error level from a 16-entry table (10 ppm .. 2 %, median ~450 ppm); alt
// count ~ floor(lambda) + Bernoulli(frac) + a small integer dispersion term;
// 2 % of cells absent, 1 % low coverage, 0.1 % of normal cells carry a
// heterozygous SNP (alt share 35-65 %); tumours add spiked SNVs at 0.1 % of
// positions with VAF in {1,2,5,20,50 %}.  RD = A+C+G+T always.
#ifndef AMPLI_SYNTH_H
#define AMPLI_SYNTH_H
#include <stdint.h>

#if defined(__HIPCC__)
#define AMPLI_HD __host__ __device__ inline
#else
#define AMPLI_HD static inline
#endif

#define AMPLI_SYNTH_ABSENT INT32_MIN

AMPLI_HD uint64_t ampli_splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

AMPLI_HD uint64_t ampli_synth_hash(uint64_t seed, uint64_t p, uint64_t s, uint64_t field)
{
    return ampli_splitmix64(seed ^ ampli_splitmix64(p * 0xD1B54A32D192ED03ull + s * 0x8CB92BA72F3D8DD7ull + field));
}

AMPLI_HD int ampli_synth_ref_base(uint64_t seed, uint64_t p)
{
    return (int)(ampli_synth_hash(seed, p, 0xFFFFFFFFull, 1) & 3);
}

// error level in parts per million for (p, nt, strand)
AMPLI_HD uint32_t ampli_synth_rate_ppm(uint64_t seed, uint64_t p, int nt, int strand)
{
    const uint32_t tab[16] = {10, 20, 50, 100, 150, 200, 300, 400, 500, 600, 800, 1000, 2000, 5000, 10000, 20000};
    // bias towards the middle of the table: average two 4-bit draws
    uint64_t h = ampli_synth_hash(seed, p, 0xFFFFFFFEull, 16 + (uint64_t)(nt * 2 + strand));
    uint32_t a = (uint32_t)(h & 15), b = (uint32_t)((h >> 4) & 15);
    return tab[(a + b + ((h >> 8) & 1)) >> 1];
}

AMPLI_HD int32_t ampli_synth_alt_count(uint32_t depth, uint32_t rate_ppm, uint64_t h)
{
    // lambda in 1e-6 units
    uint64_t lam = (uint64_t)depth * rate_ppm;
    uint32_t k = (uint32_t)(lam / 1000000ull);
    uint32_t frac = (uint32_t)(lam % 1000000ull);
    uint32_t u = (uint32_t)(h & 0xFFFFF) % 1000000u; // ~uniform [0,1e6)
    if (u < frac) k += 1;
    // dispersion: +1 (1/8), +2 (1/64), +3 (1/256), -1 (1/8)
    uint32_t d = (uint32_t)((h >> 24) & 0xFF);
    if (d < 32) k += 1;
    else if (d < 36) k += 2;
    else if (d < 37) k += 3;
    else if (d < 69 && k > 0) k -= 1;
    // larger lambda: widen roughly like sqrt(lambda)
    if (lam > 16000000ull) {
        uint32_t sd = 4;
        while ((uint64_t)sd * sd * 1000000ull < lam) sd++;
        int32_t z = (int32_t)((h >> 32) & 7) - 3; // -3..4
        int64_t kk = (int64_t)k + ((int64_t)z * sd) / 2;
        k = kk < 0 ? 0u : (uint32_t)kk;
    }
    return (int32_t)k;
}

// Writes rec[8] = {Afw,Cfw,Gfw,Tfw,Ars,Crs,Grs,Trs}; rec[0] = AMPLI_SYNTH_ABSENT for an absent cell.
// s is the GLOBAL sample index; tumour != 0 switches the sample family (different hash
// domain + spiked SNVs).  depth = nominal total depth.
AMPLI_HD void ampli_synth_record(uint64_t seed, uint64_t p, uint64_t s, int32_t depth, int tumour, int32_t *rec)
{
    const uint32_t scale_tab[16] = {512, 600, 680, 760, 840, 920, 980, 1024, 1070, 1150, 1250, 1370, 1500, 1680, 1860, 2048};
    const uint32_t eff_tab[16] = {614, 680, 740, 800, 860, 920, 970, 1024, 1024, 1080, 1130, 1190, 1250, 1310, 1370, 1434};
    const uint64_t sdom = tumour ? (s | 0x40000000ull) : s;
    uint64_t hc = ampli_synth_hash(seed, p, sdom, 2);
    if (hc % 1000 < 20) { // absent
        rec[0] = AMPLI_SYNTH_ABSENT;
        for (int i = 1; i < 8; ++i) rec[i] = 0;
        return;
    }
    uint32_t scale = scale_tab[ampli_synth_hash(seed, 0xFFFFFFFDull, sdom, 3) & 15];
    uint32_t eff = eff_tab[ampli_synth_hash(seed, p, 0xFFFFFFFCull, 4) & 15];
    uint64_t rd = ((uint64_t)depth * scale * eff) >> 20;
    rd = rd * (974 + (((hc >> 16) & 0xFF) * 100 >> 8)) >> 10;
    if ((hc >> 32) % 100 == 0) rd >>= 4; // low coverage cell
    if (rd < 2) rd = 2;
    uint32_t fwshare = 450 + (uint32_t)((hc >> 40) % 101);
    uint32_t FW = (uint32_t)(rd * fwshare / 1000), BW = (uint32_t)rd - FW;
    int ref = ampli_synth_ref_base(seed, p);
    int32_t fw[4], bw[4];
    uint32_t used_fw = 0, used_bw = 0;
    for (int nt = 0; nt < 4; ++nt) {
        if (nt == ref) { fw[nt] = bw[nt] = 0; continue; }
        uint64_t h1 = ampli_synth_hash(seed, p, sdom, 32 + (uint64_t)nt * 2);
        uint64_t h2 = ampli_synth_hash(seed, p, sdom, 33 + (uint64_t)nt * 2);
        fw[nt] = ampli_synth_alt_count(FW, ampli_synth_rate_ppm(seed, p, nt, 0), h1);
        bw[nt] = ampli_synth_alt_count(BW, ampli_synth_rate_ppm(seed, p, nt, 1), h2);
    }
    // heterozygous SNP in a normal cell / spiked SNV in a tumour
    uint64_t hv = ampli_synth_hash(seed, p, sdom, 5);
    int alt = (ref + 1 + (int)((hv >> 8) % 3)) & 3;
    if (!tumour) {
        if (hv % 1000 == 0) {
            uint32_t share = 350 + (uint32_t)((hv >> 16) % 301);
            fw[alt] += (int32_t)((uint64_t)FW * share / 1000);
            bw[alt] += (int32_t)((uint64_t)BW * share / 1000);
        }
    } else {
        uint64_t hp = ampli_synth_hash(seed, p, 0xFFFFFFFBull, 6); // position-level: is this a spiked site
        if (hp % 1000 == 0 && (hv & 1)) {
            const uint32_t vaf_tab[5] = {10, 20, 50, 200, 500}; // per mille
            uint32_t vaf = vaf_tab[(hp >> 16) % 5];
            int a2 = (ref + 1 + (int)((hp >> 24) % 3)) & 3;
            fw[a2] += (int32_t)((uint64_t)FW * vaf / 1000);
            bw[a2] += (int32_t)((uint64_t)BW * vaf / 1000);
        }
    }
    for (int nt = 0; nt < 4; ++nt) {
        if (nt == ref) continue;
        if ((uint32_t)fw[nt] > FW - used_fw) fw[nt] = (int32_t)(FW - used_fw);
        if ((uint32_t)bw[nt] > BW - used_bw) bw[nt] = (int32_t)(BW - used_bw);
        used_fw += (uint32_t)fw[nt];
        used_bw += (uint32_t)bw[nt];
    }
    fw[ref] = (int32_t)(FW - used_fw);
    bw[ref] = (int32_t)(BW - used_bw);
    for (int nt = 0; nt < 4; ++nt) { rec[nt] = fw[nt]; rec[4 + nt] = bw[nt]; }
}

#endif
