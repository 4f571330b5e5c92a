// amplisolve_amd/csrc/host/table.cpp -- positionSpecificNoise_*.txt writer (EE:2546-3043) and reader (VC:430-576).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <algorithm>
#include <sstream>
#include <thread>

#include "host.hpp"

namespace ampli {

// Thres_X cell: "-2_-2" when X is the panel reference base (EE:2670-2673), "0.01_0.01" when the estimate
// is missing ("-1_-1" in the reference's map, rewritten at EE:2680-2684), else "%f_%f" (EE:1704).
std::string format_rate_cell(uint8_t code, float r_fw, float r_bw, bool is_ref)
{
    if (is_ref) return "-2_-2";
    if (code) return "0.01_0.01";
    char buf[128];
    snprintf(buf, sizeof buf, "%f_%f", (double)r_fw, (double)r_bw);
    return buf;
}

// Germ_Max_X cell: "-" when no record qualified, else `ostream << double` (EE:2807-2849)
std::string format_germ_cell(uint8_t present, float v)
{
    if (!present) return "-";
    std::ostringstream os;
    os << (double)v;
    return os.str();
}

static const char *kHeader =
    "chrom\tposition\treference\tduplicate\tThres_A\tThres_C\tThres_G\tThres_T\tGerm_Max_A\tGerm_Max_C\tGerm_Max_G\tGerm_Max_T";

// One row per BED-walk position, duplicates repeated (EE:2575-2606).  The rows are formatted by a few threads into per-slice
// buffers and written in order (a 100 k-row table took 0.12 s on one thread -- a fifth of the whole command line at config 3).
void write_error_table(const Panel &panel, const float *rate, const uint8_t *code, const float *germ_val,
                       const uint8_t *germ_present, const std::string &path)
{
    const int64_t P = panel.P();
    const size_t n = panel.walk.size();
    int n_threads = (int)std::min<size_t>(8, std::max<size_t>(1, n / 4096));
    if (const char *e = getenv("AMPLISOLVE_THREADS")) n_threads = std::max(1, std::min(n_threads, atoi(e)));
    std::vector<std::string> part((size_t)n_threads);
    auto work = [&](int t) {
        const size_t i0 = n * (size_t)t / (size_t)n_threads, i1 = n * (size_t)(t + 1) / (size_t)n_threads;
        std::string &buf = part[(size_t)t];
        buf.reserve((i1 - i0) * 128);
        for (size_t i = i0; i < i1; ++i) {
            const uint32_t p = panel.walk[i];
            buf += panel.chroms[panel.pos_chrom[p]];
            buf += '\t';
            buf += std::to_string(panel.pos_coord[p]);
            buf += '\t';
            buf += panel.ref_base[p];
            buf += panel.dup[p] ? "\tYES" : "\tNO";
            for (int nt = 0; nt < 4; ++nt) {
                buf += '\t';
                buf += format_rate_cell(code[nt * P + p], rate[(0 * 4 + nt) * P + p], rate[(1 * 4 + nt) * P + p], panel.ref_code[p] == nt);
            }
            for (int nt = 0; nt < 4; ++nt) {
                buf += '\t';
                buf += format_germ_cell(germ_present[nt * P + p], germ_val[nt * P + p]);
            }
            buf += '\n';
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    std::ofstream out(path, std::ios::binary);
    if (!out) throw Error{AMPLI_E_INVALID, "cannot write " + path};
    out << kHeader << '\n';
    for (const std::string &b : part) out.write(b.data(), (std::streamsize)b.size());
    out.close();
    if (out.fail()) throw Error{AMPLI_E_INVALID, "cannot write " + path};
}

// germline_dir=not_available: every cell "%.4f_%.4f" of default_error, Germ_Max "-" (EE:3037-3038)
void write_error_table_default(const Panel &panel, float default_error, const std::string &path)
{
    std::ofstream out(path);
    if (!out) throw Error{AMPLI_E_INVALID, "cannot write " + path};
    out << kHeader << std::endl;
    char value[64];
    snprintf(value, sizeof value, "%.4f_%.4f", (double)default_error, (double)default_error);
    for (uint32_t p : panel.walk) {
        out << panel.chroms[panel.pos_chrom[p]] << "\t" << panel.pos_coord[p] << "\t" << panel.ref_base[p]
            << (panel.dup[p] ? "\tYES" : "\tNO");
        out << "\t" << value << "\t" << value << "\t" << value << "\t" << value << "\t-\t-\t-\t-" << '\n';
    }
}

// storeInputFile (VC:430-576): 12 whitespace-separated columns per row after the header; the FIRST row of a
// position wins (unordered_map::insert).  Thresholds become floats exactly as the call sites do it:
// sscanf("%[^_]_%[^_]") then std::stof (VC:887-890).  Also writes the by-product dummy VCF (VC:564).
void panel_from_error_table(const std::string &path, const std::string &dummy_vcf, Panel &out, std::vector<float> &thr)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) throw Error{AMPLI_E_INVALID, "Cannot open " + path};
    std::string text((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
    in.close();
    // line starts (the header is line 0)
    std::vector<size_t> ls;
    for (size_t o = 0; o < text.size();) {
        ls.push_back(o);
        const size_t e = text.find('\n', o);
        if (e == std::string::npos) break;
        text[e] = 0; // every line becomes a C string
        o = e + 1;
    }
    ls.push_back(text.size() + 1);
    const size_t n_lines = ls.size() > 1 ? ls.size() - 2 : 0; // data lines
    struct Parsed {
        char f[12][64]; // the reference scans into char[50] buffers (VC:463-474); longer cells are cut here
        float t[2][4];
        int n;
        std::string bad;
    };
    std::vector<Parsed> rows(n_lines);
    int n_threads = (int)std::min<size_t>(8, std::max<size_t>(1, n_lines / 4096));
    if (const char *e = getenv("AMPLISOLVE_THREADS")) n_threads = std::max(1, std::min(n_threads, atoi(e)));
    auto work = [&](int t) { // tokenising + strtof are the expensive part of a row and independent of every other row
        const size_t i0 = n_lines * (size_t)t / (size_t)n_threads, i1 = n_lines * (size_t)(t + 1) / (size_t)n_threads;
        for (size_t i = i0; i < i1; ++i) {
            Parsed &r = rows[i];
            for (auto &x : r.f) x[0] = 0;
            r.n = sscanf(text.c_str() + ls[i + 1], "%63s %63s %63s %63s %63s %63s %63s %63s %63s %63s %63s %63s", r.f[0], r.f[1], r.f[2], r.f[3], r.f[4],
                         r.f[5], r.f[6], r.f[7], r.f[8], r.f[9], r.f[10], r.f[11]);
            if (r.n < 1) continue;
            for (int nt = 0; nt < 4; ++nt) {
                char a[64], b[64];
                a[0] = b[0] = 0;
                sscanf(r.f[4 + nt], "%63[^_]_%63[^_]", a, b);
                char *e1 = nullptr, *e2 = nullptr;
                r.t[0][nt] = strtof(a, &e1);
                r.t[1][nt] = strtof(b, &e2);
                if ((e1 == a || e2 == b) && r.bad.empty()) r.bad = r.f[4 + nt];
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();

    std::string vcf_text;
    struct Row { float t[2][4]; };
    std::vector<Row> kept;
    for (size_t i = 0; i < n_lines; ++i) { // in file order: the FIRST row of a position wins
        const Parsed &r = rows[i];
        if (r.n < 1) continue;
        if (!dummy_vcf.empty()) { vcf_text += r.f[0]; vcf_text += '\t'; vcf_text += r.f[1]; vcf_text += "\t.\t.\t.\t.\t.\t.\n"; }
        const size_t before = out.pos_coord.size();
        const int p = out.add_position(r.f[0], atoi(r.f[1]));
        out.walk.push_back((uint32_t)p);
        if (out.pos_coord.size() == before) continue; // later rows of a duplicated position are ignored
        if (!r.bad.empty()) throw Error{AMPLI_E_INVALID, "bad threshold cell '" + r.bad + "' in " + path};
        out.set_ref((uint32_t)p, r.f[2]);
        out.dup[p] = strcmp(r.f[3], "YES") == 0; // VC:508-512
        Row k;
        for (int nt = 0; nt < 4; ++nt) {
            out.thr_text[nt].emplace_back(r.f[4 + nt]);
            out.germ_text[nt].emplace_back(r.f[8 + nt]);
            k.t[0][nt] = r.t[0][nt];
            k.t[1][nt] = r.t[1][nt];
        }
        kept.push_back(k);
    }
    if (!dummy_vcf.empty()) {
        std::ofstream vcf(dummy_vcf, std::ios::binary);
        vcf.write(vcf_text.data(), (std::streamsize)vcf_text.size());
    }
    const std::vector<Row> &rows_kept = kept;
    const int64_t P = out.P();
    thr.assign((size_t)P * 8, 0.0f);
    for (int64_t p = 0; p < P; ++p)
        for (int st = 0; st < 2; ++st)
            for (int nt = 0; nt < 4; ++nt) thr[(size_t)(st * 4 + nt) * P + p] = rows_kept[p].t[st][nt];
}

} // namespace ampli
