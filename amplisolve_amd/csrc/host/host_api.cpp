// amplisolve_amd/csrc/host/host_api.cpp -- C ABI over the host library (include/amplisolve_host.h).
#include <cstring>

#include "../../../include/amplisolve_host.h"
#include "host.hpp"

using namespace ampli;

struct ampli_host_cohort {
    Panel panel;
    Cohort cohort;
    std::string err;
};

static thread_local std::string g_err;

extern "C" const char *ampli_host_last_error(void) { return g_err.c_str(); }

static int cohort_load_impl(const char *bed_or_table, int is_error_table, const char *refbases_file, const char *fasta,
                            const char *aseq_dir, int n_threads, int keep_line_no, int shard_index, int shard_count,
                            ampli_host_cohort **out)
{
    if (!bed_or_table || !out) return AMPLI_E_INVALID;
    *out = nullptr;
    auto *h = new ampli_host_cohort();
    try {
        if (is_error_table) {
            std::vector<float> thr;
            panel_from_error_table(bed_or_table, "", h->panel, thr);
        } else {
            panel_from_bed(bed_or_table, h->panel);
            if (refbases_file && *refbases_file) panel_load_refbases_file(h->panel, refbases_file);
            else if (fasta && *fasta) panel_load_fasta(h->panel, fasta);
        }
        if (aseq_dir && *aseq_dir) cohort_load(h->panel, aseq_dir, "", n_threads, keep_line_no != 0, false, h->cohort, shard_index, shard_count);
        else h->cohort.P = h->panel.P();
    } catch (const Error &e) {
        g_err = e.msg;
        delete h;
        return e.code ? e.code : -1;
    }
    *out = h;
    return 0;
}

extern "C" int ampli_host_cohort_load(const char *bed_or_table, int is_error_table, const char *refbases_file, const char *fasta,
                                      const char *aseq_dir, int n_threads, int keep_line_no, ampli_host_cohort **out)
{
    return cohort_load_impl(bed_or_table, is_error_table, refbases_file, fasta, aseq_dir, n_threads, keep_line_no, 0, 1, out);
}

extern "C" int ampli_host_cohort_load_shard(const char *bed_or_table, int is_error_table, const char *refbases_file, const char *fasta,
                                            const char *aseq_dir, int n_threads, int keep_line_no, int32_t shard_index,
                                            int32_t shard_count, ampli_host_cohort **out)
{
    if (shard_count < 1 || shard_index < 0 || shard_index >= shard_count) { g_err = "bad shard"; return AMPLI_E_INVALID; }
    return cohort_load_impl(bed_or_table, is_error_table, refbases_file, fasta, aseq_dir, n_threads, keep_line_no, shard_index, shard_count, out);
}
extern "C" int32_t ampli_host_cohort_first_sample(const ampli_host_cohort *h) { return h->cohort.first_sample; }
extern "C" int32_t ampli_host_cohort_total_samples(const ampli_host_cohort *h) { return h->cohort.total_samples; }

extern "C" void ampli_host_cohort_free(ampli_host_cohort *h) { delete h; }
extern "C" int64_t ampli_host_cohort_P(const ampli_host_cohort *h) { return h->panel.P(); }
extern "C" int64_t ampli_host_cohort_E(const ampli_host_cohort *h) { return h->cohort.E; }
extern "C" int32_t ampli_host_cohort_S(const ampli_host_cohort *h) { return h->cohort.S(); }
extern "C" int64_t ampli_host_cohort_walk_len(const ampli_host_cohort *h) { return (int64_t)h->panel.walk.size(); }
extern "C" const int32_t *ampli_host_cohort_recs(const ampli_host_cohort *h) { return h->cohort.recs; }
extern "C" const uint32_t *ampli_host_cohort_dup_off(const ampli_host_cohort *h) { return h->cohort.dup_off.data(); }
extern "C" const uint32_t *ampli_host_cohort_ext_pos(const ampli_host_cohort *h) { return h->cohort.ext_pos.data(); }
extern "C" const int32_t *ampli_host_cohort_line_no(const ampli_host_cohort *h) { return h->cohort.line_no.empty() ? nullptr : h->cohort.line_no.data(); }
extern "C" const uint32_t *ampli_host_cohort_irregular(const ampli_host_cohort *h, int64_t *n)
{
    if (n) *n = (int64_t)h->cohort.irregular.size();
    return h->cohort.irregular.empty() ? nullptr : (const uint32_t *)h->cohort.irregular.data();
}
extern "C" const uint8_t *ampli_host_cohort_ref_code(const ampli_host_cohort *h) { return h->panel.ref_code.data(); }
extern "C" const uint8_t *ampli_host_cohort_dup_flag(const ampli_host_cohort *h) { return h->panel.dup.data(); }
extern "C" const char *ampli_host_cohort_sample_name(const ampli_host_cohort *h, int32_t s)
{
    return (s >= 0 && s < h->cohort.S()) ? h->cohort.names[s].c_str() : nullptr;
}
extern "C" void ampli_host_cohort_stats(const ampli_host_cohort *h, int64_t *lines, int64_t *offpanel, int64_t *irregular, int64_t *malformed)
{
    if (lines) *lines = h->cohort.n_lines;
    if (offpanel) *offpanel = h->cohort.n_offpanel;
    if (irregular) *irregular = h->cohort.n_irregular;
    if (malformed) *malformed = h->cohort.n_malformed;
}
extern "C" int ampli_host_position(const ampli_host_cohort *h, int64_t p, char *chrom_out, int chrom_cap, int32_t *coord)
{
    if (p < 0 || p >= h->panel.P()) return AMPLI_E_INVALID;
    const std::string &c = h->panel.chroms[h->panel.pos_chrom[p]];
    if (chrom_out && chrom_cap > 0) { strncpy(chrom_out, c.c_str(), (size_t)chrom_cap - 1); chrom_out[chrom_cap - 1] = 0; }
    if (coord) *coord = h->panel.pos_coord[p];
    return 0;
}

extern "C" int ampli_host_stream_chunks(const ampli_host_cohort *h, const char *aseq_dir, int n_threads, int keep_line_no, int64_t chunk_bytes,
                                        ampli_host_chunk_fn fn, void *user)
{
    if (!h || !aseq_dir || !fn || chunk_bytes <= 0) return AMPLI_E_INVALID;
    try {
        static_assert(sizeof(Irregular) == 16, "Irregular is four 32-bit words");
        ChunkStream cs(h->panel, list_count_files(aseq_dir, ""), n_threads, keep_line_no != 0, (size_t)chunk_bytes, 3);
        for (Chunk *c; (c = cs.next()) != nullptr;) {
            const int rc = fn(user, c->first, c->n, c->layout, c->P, c->E, c->prim, c->E ? c->ext : nullptr, c->dup_off.data(),
                              c->E ? c->ext_pos.data() : nullptr, keep_line_no ? c->line_prim.data() : nullptr,
                              (keep_line_no && c->E) ? c->line_ext.data() : nullptr, (const uint32_t *)c->irregular.data(),
                              (int64_t)c->irregular.size());
            cs.release(c);
            if (rc != 0) return rc;
        }
    } catch (const Error &e) {
        g_err = e.msg;
        return e.code ? e.code : -1;
    }
    return 0;
}

extern "C" int ampli_host_write_error_table(const ampli_host_cohort *h, const float *rate, const uint8_t *code, const float *germ_val,
                                            const uint8_t *germ_present, const char *path)
{
    try {
        write_error_table(h->panel, rate, code, germ_val, germ_present, path);
    } catch (const Error &e) {
        g_err = e.msg;
        return e.code ? e.code : -1;
    }
    return 0;
}

extern "C" int ampli_host_read_error_table_vcf(const char *path, const char *dummy_vcf, ampli_host_cohort **out, float *thr_out, int64_t thr_capacity)
{
    auto *h = new ampli_host_cohort();
    try {
        std::vector<float> thr;
        panel_from_error_table(path, dummy_vcf ? dummy_vcf : "", h->panel, thr);
        h->cohort.P = h->panel.P();
        if (thr_out) {
            if ((int64_t)thr.size() > thr_capacity) throw Error{AMPLI_E_INVALID, "thr buffer too small"};
            memcpy(thr_out, thr.data(), thr.size() * sizeof(float));
        }
    } catch (const Error &e) {
        g_err = e.msg;
        delete h;
        return e.code ? e.code : -1;
    }
    *out = h;
    return 0;
}

extern "C" int ampli_host_read_error_table(const char *path, ampli_host_cohort **out, float *thr_out, int64_t thr_capacity)
{
    return ampli_host_read_error_table_vcf(path, nullptr, out, thr_out, thr_capacity);
}

extern "C" const char *ampli_host_table_cell(const ampli_host_cohort *h, int64_t p, int32_t which)
{
    if (!h || p < 0 || p >= h->panel.P() || which < 0 || which > 8) return nullptr;
    const Panel &pn = h->panel;
    if (which == 0) return (size_t)p < pn.ref_base.size() ? pn.ref_base[(size_t)p].c_str() : nullptr;
    if (pn.cell_off[which - 1].size() != (size_t)pn.P()) return nullptr; // not loaded from a table
    return pn.table_text.data() + pn.cell_off[which - 1][(size_t)p];
}

extern "C" int ampli_host_context(const ampli_host_cohort *h, int64_t p, char sub, char *down, char *up, int32_t cap)
{
    if (!h || !down || !up || cap <= 0 || p < 0 || p >= h->panel.P()) return AMPLI_E_INVALID;
    const Panel &pn = h->panel;
    const std::string &chrom = pn.chroms[(size_t)pn.pos_chrom[(size_t)p]];
    const std::string d = kmer_down(pn, chrom, pn.pos_coord[(size_t)p]), u = kmer_up(pn, chrom, pn.pos_coord[(size_t)p]);
    if ((int64_t)d.size() >= cap || (int64_t)u.size() >= cap) return AMPLI_E_INVALID;
    memcpy(down, d.c_str(), d.size() + 1);
    memcpy(up, u.c_str(), u.size() + 1);
    return homopolymer_test(d, u, sub);
}

extern "C" int ampli_host_run_error_estimation(const char *panel_design, const char *reference_genome, const char *germline_dir,
                                               const char *C_value, const char *coverage_cutoff, const char *default_error,
                                               const char *output_dir, const char *refbases_file)
{
    EeArgs a;
    a.panel_design = panel_design; a.reference_genome = reference_genome ? reference_genome : ""; a.germline_dir = germline_dir;
    a.C_value = C_value; a.coverage_cutoff = coverage_cutoff; a.default_error = default_error; a.output_dir = output_dir;
    if (refbases_file) a.refbases_file = refbases_file;
    return run_error_estimation(a);
}

extern "C" int ampli_host_run_variant_calling(const char *error_file, const char *tumour_dir, const char *output_dir,
                                              const char *coverage_cutoff, const char *p_value)
{
    VcArgs a;
    a.error_file = error_file; a.tumour_dir = tumour_dir; a.output_dir = output_dir; a.coverage_cutoff = coverage_cutoff; a.p_value = p_value;
    return run_variant_calling(a);
}

static bool shard_ok(const ampli_host_shard *sh)
{
    return sh && sh->count >= 1 && sh->index >= 0 && sh->index < sh->count &&
           (sh->count == 1 || (sh->ee_buffers && sh->ee_exchange && sh->ee_gather && sh->or_flags && sh->rows_before && sh->barrier));
}

extern "C" int ampli_host_run_error_estimation_sharded(const char *panel_design, const char *reference_genome, const char *germline_dir,
                                                       const char *C_value, const char *coverage_cutoff, const char *default_error,
                                                       const char *output_dir, const char *refbases_file, const ampli_host_shard *shard)
{
    if (!shard_ok(shard)) { g_err = "bad shard description"; return AMPLI_E_INVALID; }
    EeArgs a;
    a.panel_design = panel_design; a.reference_genome = reference_genome ? reference_genome : ""; a.germline_dir = germline_dir;
    a.C_value = C_value; a.coverage_cutoff = coverage_cutoff; a.default_error = default_error; a.output_dir = output_dir;
    if (refbases_file) a.refbases_file = refbases_file;
    a.shard = shard;
    return run_error_estimation(a);
}

extern "C" int ampli_host_run_variant_calling_sharded(const char *error_file, const char *tumour_dir, const char *output_dir,
                                                      const char *coverage_cutoff, const char *p_value, const ampli_host_shard *shard)
{
    if (!shard_ok(shard)) { g_err = "bad shard description"; return AMPLI_E_INVALID; }
    VcArgs a;
    a.error_file = error_file; a.tumour_dir = tumour_dir; a.output_dir = output_dir; a.coverage_cutoff = coverage_cutoff; a.p_value = p_value;
    a.shard = shard;
    return run_variant_calling(a);
}

extern "C" int ampli_host_compute_counts(const char *vcf, const char *bam, const char *out_dir, int32_t threads, int32_t mbq, int32_t mrq, int32_t mdc,
                                         int64_t *stats)
{
    if (!vcf || !bam || !out_dir) { g_err = "vcf, bam and out_dir are required"; return AMPLI_E_INVALID; }
    CcArgs a;
    a.vcf = vcf; a.bam = bam; a.out_dir = out_dir; a.threads = threads; a.mbq = mbq; a.mrq = mrq; a.mdc = mdc; a.stats = stats;
    std::string err;
    a.error = &err;
    const int rc = run_compute_counts(a);
    if (rc) g_err = err;
    return rc;
}

extern "C" int ampli_host_bam_scan(const char *bam, int32_t threads, int64_t *stats)
{
    if (!bam || !stats) { g_err = "bam and stats are required"; return AMPLI_E_INVALID; }
    try {
        bam_scan(bam, threads, stats);
        return 0;
    } catch (const Error &e) {
        g_err = e.msg;
        return e.code ? e.code : -1;
    }
}

extern "C" double ampli_host_fisher(int a, int b, int c, int d) { return fisher_two_sided(a, b, c, d); }
extern "C" double ampli_host_fisher_direct(int a, int b, int c, int d) { return fisher_two_sided_direct(a, b, c, d); }

extern "C" double ampli_host_guard_score(int32_t k, int32_t rd, float err, int32_t *ge5, int32_t *lt20)
{
    const long double q = score_reference_sequence(k, rd, err);
    if (ge5) *ge5 = q >= 5 ? 1 : 0;   // the comparisons of VC:898 / VC:1023, in long double like the reference's
    if (lt20) *lt20 = q < 20 ? 1 : 0;
    return (double)q;
}

extern "C" int ampli_host_sample_order(const char *dir, char *out, int64_t cap)
{
    try {
        auto v = list_count_files(dir, "");
        std::string s;
        for (auto &f : v) s += f.second + "\n";
        if ((int64_t)s.size() + 1 > cap) return AMPLI_E_INVALID;
        memcpy(out, s.c_str(), s.size() + 1);
        return (int)v.size();
    } catch (const Error &e) {
        g_err = e.msg;
        return e.code ? e.code : -1;
    }
}
