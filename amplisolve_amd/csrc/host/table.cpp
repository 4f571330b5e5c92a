// amplisolve_amd/csrc/host/table.cpp -- positionSpecificNoise_*.txt writer (EE:2546-3043) and reader (VC:430-576).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

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

// One row per BED-walk position, duplicates repeated (EE:2575-2606).
void write_error_table(const Panel &panel, const float *rate, const uint8_t *code, const float *germ_val,
                       const uint8_t *germ_present, const std::string &path)
{
    std::ofstream out(path);
    if (!out) throw Error{AMPLI_E_INVALID, "cannot write " + path};
    const int64_t P = panel.P();
    out << kHeader << std::endl;
    std::string row;
    for (uint32_t p : panel.walk) {
        row.clear();
        row += panel.chroms[panel.pos_chrom[p]];
        row += '\t';
        row += std::to_string(panel.pos_coord[p]);
        row += '\t';
        row += panel.ref_base[p];
        row += panel.dup[p] ? "\tYES" : "\tNO";
        for (int nt = 0; nt < 4; ++nt) {
            row += '\t';
            row += format_rate_cell(code[nt * P + p], rate[(0 * 4 + nt) * P + p], rate[(1 * 4 + nt) * P + p],
                                    panel.ref_code[p] == nt);
        }
        for (int nt = 0; nt < 4; ++nt) {
            row += '\t';
            row += format_germ_cell(germ_present[nt * P + p], germ_val[nt * P + p]);
        }
        out << row << '\n';
    }
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
    std::ifstream in(path);
    if (!in) throw Error{AMPLI_E_INVALID, "Cannot open " + path};
    std::ofstream vcf;
    if (!dummy_vcf.empty()) vcf.open(dummy_vcf);
    std::string line;
    std::getline(in, line); // header
    struct Row { float t[2][4]; };
    std::vector<Row> rows;
    while (std::getline(in, line)) {
        char f[12][1024];
        for (auto &x : f) x[0] = 0;
        const int n = sscanf(line.c_str(), "%1000s %1000s %1000s %1000s %1000s %1000s %1000s %1000s %1000s %1000s %1000s %1000s", f[0],
                             f[1], f[2], f[3], f[4], f[5], f[6], f[7], f[8], f[9], f[10], f[11]);
        if (n < 1) continue;
        if (vcf.is_open()) vcf << f[0] << "\t" << f[1] << "\t.\t.\t.\t.\t.\t." << std::endl;
        const size_t before = out.pos_coord.size();
        const int p = out.add_position(f[0], atoi(f[1]));
        out.walk.push_back((uint32_t)p);
        if (out.pos_coord.size() == before) continue; // later rows of a duplicated position are ignored
        out.set_ref((uint32_t)p, f[2]);
        out.dup[p] = strcmp(f[3], "YES") == 0; // VC:508-512
        Row r;
        for (int nt = 0; nt < 4; ++nt) {
            out.thr_text[nt].emplace_back(f[4 + nt]);
            out.germ_text[nt].emplace_back(f[8 + nt]);
            char a[1024], b[1024];
            a[0] = b[0] = 0;
            sscanf(f[4 + nt], "%[^_]_%[^_]", a, b);
            char *e1 = nullptr, *e2 = nullptr;
            r.t[0][nt] = strtof(a, &e1);
            r.t[1][nt] = strtof(b, &e2);
            if (e1 == a || e2 == b) throw Error{AMPLI_E_INVALID, "bad threshold cell '" + std::string(f[4 + nt]) + "' in " + path};
        }
        rows.push_back(r);
    }
    const int64_t P = out.P();
    thr.assign((size_t)P * 8, 0.0f);
    for (int64_t p = 0; p < P; ++p)
        for (int st = 0; st < 2; ++st)
            for (int nt = 0; nt < 4; ++nt) thr[(size_t)(st * 4 + nt) * P + p] = rows[p].t[st][nt];
}

} // namespace ampli
