// amplisolve_amd/csrc/host/table.cpp -- positionSpecificNoise_*.txt writer (EE:2546-3043) and reader (VC:430-576).
#include <cmath>
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
    char buf[64];
    snprintf(buf, sizeof buf, "%g", (double)v); // what `ostream << double` prints with its default flags: "%.6g" (libstdc++ formats through vsnprintf)
    return buf;
}

static const char *kHeader =
    "chrom\tposition\treference\tduplicate\tThres_A\tThres_C\tThres_G\tThres_T\tGerm_Max_A\tGerm_Max_C\tGerm_Max_G\tGerm_Max_T";

namespace {
// sprintf("%f", (double)r) by hand: glibc prints round-half-even(r * 10^6) / 10^6 of the exact binary value; r = m * 2^e
// with m < 2^24, so m * 10^6 < 2^44 and the rounding is exact in 64-bit integers (the arithmetic of ampli_text_roundtrip,
// csrc/ampli_math.h).  Returns the end of the text; values it does not cover (|r| >= 2^20, inf, nan) go through snprintf.
char *put_f6(char *p, float r)
{
    uint32_t bits;
    memcpy(&bits, &r, 4);
    const uint32_t ex = (bits >> 23) & 0xFF;
    uint32_t man = bits & 0x7FFFFFu;
    if (ex == 0xFF || ex >= 127 + 20) return p + sprintf(p, "%f", (double)r);
    if (bits & 0x80000000u) *p++ = '-';
    int e2;
    if (ex == 0) {
        e2 = -149;
    } else {
        man |= 0x800000u;
        e2 = (int)ex - 150;
    }
    // |r| = man * 2^e2 with e2 in [-149, -4]
    const uint64_t A = (uint64_t)man * 1000000ull; // < 2^44
    const int k = -e2;
    uint64_t N;
    if (k >= 45) {
        N = 0;
    } else {
        const uint64_t q = A >> k, rem = A & ((1ull << k) - 1), half = 1ull << (k - 1);
        N = q + ((rem > half) || (rem == half && (q & 1)));
    }
    const uint64_t ip = N / 1000000ull;
    uint32_t fr = (uint32_t)(N % 1000000ull);
    char tmp[24];
    int n = 0;
    uint64_t v = ip;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n) *p++ = tmp[--n];
    *p++ = '.';
    for (int i = 5; i >= 0; --i) { p[i] = (char)('0' + fr % 10); fr /= 10; }
    return p + 6;
}

char *put_str(char *p, const char *s_, size_t n)
{
    memcpy(p, s_, n);
    return p + n;
}

char *put_int(char *p, int v)
{
    if (v < 0) return p + sprintf(p, "%d", v);
    char tmp[16];
    int n = 0;
    unsigned u = (unsigned)v;
    do { tmp[n++] = (char)('0' + u % 10); u /= 10; } while (u);
    while (n) *p++ = tmp[--n];
    return p;
}
} // namespace

// One row per BED-walk position, duplicates repeated (EE:2575-2606).  The rows are formatted by a few threads into per-slice
// buffers, cell by cell without temporaries ("%f" by exact integer arithmetic, put_f6), and written in order: a 100 k-row
// table took 0.12 s on one thread in round 1, 0.04 s on 8 threads of snprintf + std::string in round 3.
void write_error_table(const Panel &panel, const float *rate, const uint8_t *code, const float *germ_val,
                       const uint8_t *germ_present, const std::string &path)
{
    const int64_t P = panel.P();
    const size_t n = panel.walk.size();
    int n_threads = (int)std::min<size_t>(std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())), std::max<size_t>(1, n / 4096));
    if (const char *e = getenv("AMPLISOLVE_THREADS")) n_threads = std::max(1, std::min(n_threads, atoi(e)));
    size_t longest_chrom = 0, longest_ref = 0;
    for (auto &c : panel.chroms) longest_chrom = std::max(longest_chrom, c.size());
    for (auto &r : panel.ref_base) longest_ref = std::max(longest_ref, r.size());
    // a row at its longest: chrom, position (<= 11), reference, YES/NO, 4 x "%f_%f" (each %f <= 48 characters for a finite
    // float), 4 x "%g" (<= 13), 12 separators; a typical row is ~110 characters, so the slice buffer starts at 160 per row
    // (uninitialised memory: zero-filling tens of MB cost more than the formatting) and grows if a row might not fit
    const size_t row_cap = longest_chrom + longest_ref + 11 + 3 + 4 * (2 * 48 + 1) + 4 * 16 + 16;
    struct Slice { char *base = nullptr; size_t len = 0; ~Slice() { free(base); } };
    std::vector<Slice> part((size_t)n_threads);
    auto work = [&](int t) {
        const size_t i0 = n * (size_t)t / (size_t)n_threads, i1 = n * (size_t)(t + 1) / (size_t)n_threads;
        size_t cap = (i1 - i0) * (longest_chrom + longest_ref + 160) + row_cap;
        char *base = (char *)malloc(cap);
        if (!base) return; // reported below: a slice without a buffer
        char *q = base;
        for (size_t i = i0; i < i1; ++i) {
            if ((size_t)(q - base) + row_cap > cap) {
                const size_t used = (size_t)(q - base);
                cap = cap + cap / 2 + row_cap;
                char *nb = (char *)realloc(base, cap);
                if (!nb) { free(base); return; }
                base = nb;
                q = base + used;
            }
            const uint32_t p = panel.walk[i];
            const std::string &chrom = panel.chroms[panel.pos_chrom[p]];
            q = put_str(q, chrom.data(), chrom.size());
            *q++ = '\t';
            q = put_int(q, panel.pos_coord[p]);
            *q++ = '\t';
            q = put_str(q, panel.ref_base[p].data(), panel.ref_base[p].size());
            q = panel.dup[p] ? put_str(q, "\tYES", 4) : put_str(q, "\tNO", 3);
            for (int nt = 0; nt < 4; ++nt) {
                *q++ = '\t';
                // Thres_X cell: "-2_-2" when X is the panel reference base (EE:2670-2673), "0.01_0.01" when the estimate is
                // missing (EE:2680-2684), else "%f_%f" (EE:1704) -- format_rate_cell, without the temporaries
                if (panel.ref_code[p] == nt) q = put_str(q, "-2_-2", 5);
                else if (code[nt * P + p]) q = put_str(q, "0.01_0.01", 9);
                else {
                    q = put_f6(q, rate[(0 * 4 + nt) * P + p]);
                    *q++ = '_';
                    q = put_f6(q, rate[(1 * 4 + nt) * P + p]);
                }
            }
            for (int nt = 0; nt < 4; ++nt) {
                *q++ = '\t';
                // Germ_Max_X cell: "-" when no record qualified, else `ostream << double` = "%g" (EE:2807-2849)
                if (!germ_present[nt * P + p]) *q++ = '-';
                else {
                    const float v = germ_val[nt * P + p];
                    if (v == 0.0f && !std::signbit(v)) *q++ = '0';
                    else q += sprintf(q, "%g", (double)v);
                }
            }
            *q++ = '\n';
        }
        part[(size_t)t].base = base;
        part[(size_t)t].len = (size_t)(q - base);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();
    std::ofstream out(path, std::ios::binary);
    if (!out) throw Error{AMPLI_E_INVALID, "cannot write " + path};
    out << kHeader << '\n';
    for (size_t t = 0; t < part.size(); ++t) {
        if (!part[t].base && n * (t + 1) / part.size() > n * t / part.size()) throw Error{AMPLI_E_NOMEM, "out of memory formatting " + path};
        out.write(part[t].base, (std::streamsize)part[t].len);
    }
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
    const double tt0 = PhaseClock::now();
    std::ifstream in(path, std::ios::binary);
    if (!in) throw Error{AMPLI_E_INVALID, "Cannot open " + path};
    in.seekg(0, std::ios::end);
    const std::streamoff sz = in.tellg();
    in.seekg(0, std::ios::beg);
    std::string &text = out.table_text; // the panel keeps the text: every cell below is a C string inside it
    text.resize(sz > 0 ? (size_t)sz : 0);
    if (sz > 0) in.read(&text[0], sz);
    in.close();
    text.push_back('\0');
    // line starts (the header is line 0)
    std::vector<size_t> ls;
    for (size_t o = 0; o + 1 < text.size();) {
        ls.push_back(o);
        const void *e = memchr(text.data() + o, '\n', text.size() - 1 - o);
        if (!e) break;
        const size_t eo = (size_t)((const char *)e - text.data());
        text[eo] = 0; // every line becomes a C string
        o = eo + 1;
    }
    ls.push_back(text.size());
    const size_t n_lines = ls.size() > 1 ? ls.size() - 2 : 0; // data lines
    struct Parsed {
        uint32_t f[12]; // offsets of the row's cells in `text`, each NUL-terminated in place (0 = absent: offset 0 is the header)
        float t[2][4];
        int32_t n, coord;
        int32_t bad; // 1 + index of a threshold cell that is not two numbers, 0 = fine
    };
    const double tt1 = PhaseClock::now();
    std::vector<Parsed> rows(n_lines);
    int n_threads = (int)std::min<size_t>(std::min<size_t>(16, std::max(1u, std::thread::hardware_concurrency())), std::max<size_t>(1, n_lines / 4096));
    if (const char *e = getenv("AMPLISOLVE_THREADS")) n_threads = std::max(1, std::min(n_threads, atoi(e)));
    auto is_space = [](char ch) { return ch == ' ' || ch == '\t' || ch == '\r' || ch == '\n' || ch == '\v' || ch == '\f'; };
    auto work = [&](int t) { // tokenising + strtof are the expensive part of a row and independent of every other row
        const size_t i0 = n_lines * (size_t)t / (size_t)n_threads, i1 = n_lines * (size_t)(t + 1) / (size_t)n_threads;
        for (size_t i = i0; i < i1; ++i) {
            Parsed &r = rows[i];
            memset(&r, 0, sizeof r);
            // the twelve "%s" of VC:476 by hand: runs of non-space characters; the reference scans into char[50] buffers
            // (VC:463-474), a longer cell is cut at 63 characters here
            char *q = &text[ls[i + 1]];
            while (r.n < 12) {
                while (*q && is_space(*q)) ++q;
                if (!*q) break;
                char *b = q;
                while (*q && !is_space(*q)) ++q;
                const bool more = *q != 0;
                if (q - b > 63) b[63] = 0;
                *q = 0;
                r.f[r.n++] = (uint32_t)(b - text.data());
                if (more) ++q;
            }
            if (r.n < 1) continue;
            r.coord = r.n > 1 ? atoi(text.data() + r.f[1]) : 0;
            for (int nt = 0; nt < 4; ++nt) {
                // sscanf("%[^_]_%[^_]") of VC:886: a = the characters before the first '_' (none: nothing is converted at
                // all), then the '_', then b = the characters up to the next '_'
                char a[64], b[64];
                a[0] = b[0] = 0;
                const char *c = r.n > 4 + nt ? text.data() + r.f[4 + nt] : "";
                int k = 0;
                while (c[k] && c[k] != '_') { a[k] = c[k]; ++k; }
                a[k] = 0;
                if (k > 0 && c[k] == '_') {
                    const char *d = c + k + 1;
                    int m = 0;
                    while (d[m] && d[m] != '_') { b[m] = d[m]; ++m; }
                    b[m] = 0;
                }
                char *e1 = nullptr, *e2 = nullptr;
                r.t[0][nt] = strtof(a, &e1);
                r.t[1][nt] = strtof(b, &e2);
                if ((e1 == a || e2 == b) && !r.bad) r.bad = 1 + 4 + nt;
            }
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &x : th) x.join();

    const double tt2 = PhaseClock::now();
    // the by-product dummy VCF (VC:437-441, 571) beside the serial pass below
    std::thread vcf_writer;
    if (!dummy_vcf.empty())
        vcf_writer = std::thread([&] {
            std::string vcf_text;
            vcf_text.reserve(n_lines * 32);
            for (size_t i = 0; i < n_lines; ++i) {
                const Parsed &r = rows[i];
                if (r.n < 1) continue;
                vcf_text += text.data() + r.f[0];
                vcf_text += '\t';
                if (r.n > 1) vcf_text += text.data() + r.f[1];
                vcf_text += "\t.\t.\t.\t.\t.\t.\n";
            }
            std::ofstream vcf(dummy_vcf, std::ios::binary);
            vcf.write(vcf_text.data(), (std::streamsize)vcf_text.size());
        });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{vcf_writer};

    std::vector<uint32_t> kept; // row index of each unique position
    kept.reserve(n_lines);
    out.walk.reserve(n_lines);
    const char *empty = text.data() + text.size() - 1; // the terminating NUL: an absent cell reads as ""
    auto cell = [&](const Parsed &r, int k) { return r.n > k ? text.data() + r.f[k] : empty; };
    std::string last_chrom;
    int last_cid = -1;
    for (size_t i = 0; i < n_lines; ++i) { // in file order: the FIRST row of a position wins
        const Parsed &r = rows[i];
        if (r.n < 1) continue;
        const char *chrom = cell(r, 0);
        if (last_cid < 0 || last_chrom != chrom) { // rows of a chromosome come together: one string look-up per run
            last_chrom = chrom;
            auto c = out.chrom_id.find(last_chrom);
            if (c == out.chrom_id.end()) {
                last_cid = (int)out.chroms.size();
                out.chroms.push_back(last_chrom);
                out.chrom_id.emplace(last_chrom, last_cid);
            } else {
                last_cid = c->second;
            }
        }
        const size_t before = out.pos_coord.size();
        const int p = out.add_position_id(last_cid, r.coord);
        out.walk.push_back((uint32_t)p);
        if (out.pos_coord.size() == before) continue; // later rows of a duplicated position are ignored
        if (r.bad) throw Error{AMPLI_E_INVALID, std::string("bad threshold cell '") + cell(r, r.bad - 1) + "' in " + path};
        out.set_ref((uint32_t)p, cell(r, 2));
        out.dup[p] = strcmp(cell(r, 3), "YES") == 0; // VC:508-512
        kept.push_back((uint32_t)i);
    }
    const double tt3 = PhaseClock::now();
    const int64_t P = out.P();
    for (int k = 0; k < 8; ++k) out.cell_off[k].resize((size_t)P);
    thr.assign((size_t)P * 8, 0.0f);
    for (int64_t p = 0; p < P; ++p) {
        const Parsed &r = rows[kept[(size_t)p]];
        for (int k = 0; k < 8; ++k) out.cell_off[k][(size_t)p] = r.n > 4 + k ? r.f[4 + k] : (uint32_t)(text.size() - 1);
        for (int st = 0; st < 2; ++st)
            for (int nt = 0; nt < 4; ++nt) thr[(size_t)(st * 4 + nt) * P + p] = r.t[st][nt];
    }
    PhaseClock::add("table:read_file+lines", tt1 - tt0, false);
    PhaseClock::add("table:tokenize", tt2 - tt1, false);
    PhaseClock::add("table:index", tt3 - tt2, false);
    PhaseClock::add("table:cells", PhaseClock::now() - tt3, false);
}

} // namespace ampli
