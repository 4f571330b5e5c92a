// amplisolve_amd/csrc/host/aseq.cpp -- .PILEUP.ASEQ ingest: file list, visit order, parallel parse, SoA pack.
#include <fcntl.h>
#include <glob.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <mutex>
#include <thread>

#include "host.hpp"
#include "hip_loader.hpp"

namespace ampli {

Cohort::~Cohort()
{
    if (!recs) return;
    const HipApi *api = pinned ? hip_api() : nullptr;
    if (pinned && api) api->pinned_free(recs);
    else free(recs);
}

// `ls dir/*.ASEQ > list` (EE:552-559): the shell expands the glob in collation order, ls sorts its
// operands the same way.  storeCountList (EE:794-841) then fills an unordered_map {listed path -> name}
// and every later loop walks that map -- its iteration order IS the sample visit order (EE:1081, VC:672),
// so the very same container type is used here.
std::vector<std::pair<std::string, std::string>> list_count_files(const std::string &dir, const std::string &list_file)
{
    glob_t g;
    std::memset(&g, 0, sizeof g);
    const std::string pattern = dir + "/*.ASEQ";
    const int rc = glob(pattern.c_str(), 0, nullptr, &g);
    std::vector<std::string> listed;
    if (rc == 0)
        for (size_t i = 0; i < g.gl_pathc; ++i) listed.emplace_back(g.gl_pathv[i]);
    globfree(&g);
    if (listed.empty()) throw Error{AMPLI_E_INVALID, "no *.ASEQ files in " + dir};
    if (!list_file.empty()) {
        std::ofstream o(list_file);
        for (auto &s : listed) o << s << "\n";
    }
    // AMPLISOLVE_LIST_DIR_AS=<literal>: key the map as if the directory had been given as <literal>.  The visit
    // order is a function of the hash of the listed path strings, so this reproduces, bit for bit, outputs that
    // were generated from the same files under another directory name (another machine, another mount point).
    std::string alias = dir;
    if (const char *e = getenv("AMPLISOLVE_LIST_DIR_AS")) alias = e;
    std::unordered_map<std::string, std::string> Hash;
    std::unordered_map<std::string, std::string> real_of;
    const size_t size_DIR_name = dir.size();
    for (auto &line : listed) {
        // EE:829-832: name = listed path minus "<dir>/" minus the 12 characters of ".PILEUP.ASEQ"
        std::string name;
        if (line.size() >= size_DIR_name + 13) name = line.substr(size_DIR_name + 1, line.size() - size_DIR_name - 13);
        const std::string key = alias + line.substr(size_DIR_name);
        Hash.insert(std::make_pair(key, name));
        real_of.emplace(key, line);
    }
    std::vector<std::pair<std::string, std::string>> out;
    for (auto it = Hash.begin(); it != Hash.end(); ++it) out.emplace_back(real_of[it->first], it->second);
    return out;
}

namespace {

struct Extra {
    uint32_t p, k;
    int32_t line;
    int32_t rec[8];
};

struct FileResult {
    std::vector<int32_t> main;   // [P][8]
    std::vector<int32_t> line;   // [P]
    std::vector<Extra> extras;
    int64_t n_lines = 0, n_off = 0, n_irregular = 0, n_malformed = 0;
    std::string error;
};

inline const char *skip_ws(const char *p, const char *e)
{
    while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
    return p;
}
inline const char *skip_tok(const char *p, const char *e)
{
    while (p < e && *p != ' ' && *p != '\t' && *p != '\r') ++p;
    return p;
}
inline bool parse_int(const char *&p, const char *e, int64_t &v)
{
    p = skip_ws(p, e);
    if (p >= e) return false;
    bool neg = false;
    if (*p == '-' || *p == '+') { neg = *p == '-'; ++p; }
    if (p >= e || *p < '0' || *p > '9') return false;
    int64_t x = 0;
    while (p < e && *p >= '0' && *p <= '9') { x = x * 10 + (*p - '0'); ++p; }
    v = neg ? -x : x;
    return true;
}

// One file.  Columns: chr pos dbsnp MAF ref alt A C G T RD Ars Crs Grs Trs (EE:1149, VC:752); the first line
// is the header (EE:1113, VC:721).  fw = X - Xrs (EE:1155-1158).
void parse_file(const Panel &panel, const std::string &path, bool keep_line, FileResult &out)
{
    const int64_t P = panel.P();
    out.main.assign((size_t)P * 8, 0);
    for (int64_t p = 0; p < P; ++p) out.main[(size_t)p * 8] = AMPLI_ABSENT;
    if (keep_line) out.line.assign((size_t)P, -1);
    std::vector<uint32_t> occ((size_t)P, 0); // occurrences so far of each position in THIS file

    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) { out.error = "Cannot open " + path; return; }
    struct stat st;
    fstat(fd, &st);
    const size_t len = (size_t)st.st_size;
    const char *base = len ? (const char *)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    if (len && base == MAP_FAILED) { close(fd); out.error = "mmap failed for " + path; return; }
    const char *cur = base, *end = base + len;
    // header
    while (cur < end && *cur != '\n') ++cur;
    if (cur < end) ++cur;

    std::string last_chrom;
    int last_cid = -1;
    int64_t prev_p = -2;
    int32_t line_idx = 0;
    while (cur < end) {
        const char *le = (const char *)memchr(cur, '\n', (size_t)(end - cur));
        if (!le) le = end;
        const char *p = skip_ws(cur, le);
        if (p < le) {
            ++out.n_lines;
            const char *c0 = p, *c1 = skip_tok(p, le);
            p = c1;
            int64_t pos, v[9];
            bool ok = parse_int(p, le, pos);
            for (int i = 0; ok && i < 4; ++i) { p = skip_ws(p, le); const char *q = skip_tok(p, le); ok = q > p; p = q; }
            for (int i = 0; ok && i < 9; ++i) ok = parse_int(p, le, v[i]);
            if (!ok) {
                ++out.n_malformed;
            } else {
                const size_t clen = (size_t)(c1 - c0);
                if (last_cid < 0 || clen != last_chrom.size() || memcmp(c0, last_chrom.data(), clen) != 0) {
                    last_chrom.assign(c0, clen);
                    auto it = panel.chrom_id.find(last_chrom);
                    last_cid = it == panel.chrom_id.end() ? -1 : it->second;
                    if (last_cid < 0) last_chrom.clear();
                }
                int64_t pi = -1;
                if (last_cid >= 0) {
                    // files follow the panel order: try the successor of the previous hit first
                    const int64_t nx = prev_p + 1;
                    if (nx >= 0 && nx < P && panel.pos_coord[nx] == pos && panel.pos_chrom[nx] == last_cid) pi = nx;
                    else {
                        auto it = panel.index.find(((uint64_t)(uint32_t)last_cid << 32) | (uint32_t)pos);
                        if (it != panel.index.end()) pi = it->second;
                    }
                }
                if (pi < 0) {
                    ++out.n_off;
                } else {
                    prev_p = pi;
                    const int64_t A = v[0], C = v[1], G = v[2], T = v[3], RD = v[4];
                    int32_t rec[8] = {(int32_t)(A - v[5]), (int32_t)(C - v[6]), (int32_t)(G - v[7]), (int32_t)(T - v[8]),
                                      (int32_t)v[5], (int32_t)v[6], (int32_t)v[7], (int32_t)v[8]};
                    if (A + C + G + T != RD) { // EE:1178-1181, VC:762-765
                        ++out.n_irregular;
                    }
                    const uint32_t k = occ[pi]++;
                    if (k == 0) {
                        memcpy(&out.main[(size_t)pi * 8], rec, sizeof rec);
                        if (keep_line) out.line[(size_t)pi] = line_idx;
                    } else {
                        Extra e;
                        e.p = (uint32_t)pi; e.k = k; e.line = line_idx;
                        memcpy(e.rec, rec, sizeof rec);
                        out.extras.push_back(e);
                    }
                }
            }
            ++line_idx;
        }
        cur = le < end ? le + 1 : end;
    }
    if (base) munmap((void *)base, len);
    close(fd);
}

} // namespace

void cohort_load(const Panel &panel, const std::string &dir, const std::string &list_file, int n_threads, bool keep_line_no,
                 bool print_irregular, Cohort &out, int shard_index, int shard_count)
{
    auto files = list_count_files(dir, list_file);
    out.total_samples = (int)files.size();
    out.first_sample = 0;
    if (shard_count > 1) { // contiguous range of the visit order; earlier shards take the remainder (dist.py::shard_range)
        const int n = (int)files.size(), base = n / shard_count, rem = n % shard_count;
        const int lo = shard_index * base + std::min(shard_index, rem), hi = lo + base + (shard_index < rem ? 1 : 0);
        files = decltype(files)(files.begin() + lo, files.begin() + hi);
        out.first_sample = lo;
    }
    const int S = (int)files.size();
    const int64_t P = panel.P();
    out.paths.clear(); out.names.clear();
    for (auto &f : files) { out.paths.push_back(f.first); out.names.push_back(f.second); }
    out.P = P;

    std::vector<FileResult> res((size_t)S);
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads > S) n_threads = S;
    if (n_threads < 1) n_threads = 1;
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&] {
            for (int s; (s = next.fetch_add(1)) < S;) parse_file(panel, out.paths[s], keep_line_no, res[s]);
        });
    for (auto &t : th) t.join();
    for (int s = 0; s < S; ++s)
        if (!res[s].error.empty()) throw Error{AMPLI_E_INVALID, res[s].error};
    if (print_irregular) // the reference's own message, once per offending line (EE:1178-1181, VC:762-765)
        for (int s = 0; s < S; ++s)
            for (int64_t i = 0; i < res[s].n_irregular; ++i) std::cout << "malakia paizei edo" << std::endl;

    // extra occurrences: slot layout from the largest multiplicity seen in any file
    std::vector<uint32_t> mult((size_t)P, 0);
    for (auto &r : res)
        for (auto &e : r.extras)
            if (e.k > mult[e.p]) mult[e.p] = e.k;
    out.dup_off.assign((size_t)P + 1, 0);
    for (int64_t p = 0; p < P; ++p) out.dup_off[p + 1] = out.dup_off[p] + mult[p];
    out.E = out.dup_off[P];
    out.ext_pos.assign((size_t)out.E, 0);
    for (int64_t p = 0; p < P; ++p)
        for (uint32_t e = out.dup_off[p]; e < out.dup_off[p + 1]; ++e) out.ext_pos[e] = (uint32_t)p;

    const int64_t R = P + out.E;
    const size_t bytes = (size_t)S * R * 8 * sizeof(int32_t);
    const HipApi *api = hip_api();
    void *mem = nullptr;
    if (api && api->device_count() > 0 && api->pinned_alloc(bytes ? bytes : 32, &mem) == AMPLI_OK) out.pinned = true;
    else { mem = aligned_alloc(256, (bytes + 255) / 256 * 256 + 256); out.pinned = false; }
    if (!mem) throw Error{AMPLI_E_NOMEM, "cannot allocate the record array"};
    out.recs = (int32_t *)mem;
    if (keep_line_no) out.line_no.assign((size_t)S * R, -1);
    for (int s = 0; s < S; ++s) {
        int32_t *dst = out.recs + (size_t)s * R * 8;
        memcpy(dst, res[s].main.data(), (size_t)P * 8 * sizeof(int32_t));
        for (int64_t e = 0; e < out.E; ++e) {
            int32_t *x = dst + (size_t)(P + e) * 8;
            x[0] = AMPLI_ABSENT;
            for (int j = 1; j < 8; ++j) x[j] = 0;
        }
        if (keep_line_no) memcpy(&out.line_no[(size_t)s * R], res[s].line.data(), (size_t)P * sizeof(int32_t));
        for (auto &e : res[s].extras) {
            const size_t slot = (size_t)P + out.dup_off[e.p] + (e.k - 1);
            memcpy(dst + slot * 8, e.rec, sizeof e.rec);
            if (keep_line_no) out.line_no[(size_t)s * R + slot] = e.line;
        }
        out.n_lines += res[s].n_lines; out.n_offpanel += res[s].n_off;
        out.n_irregular += res[s].n_irregular; out.n_malformed += res[s].n_malformed;
        FileResult().main.swap(res[s].main); // release as we go
    }
}

} // namespace ampli
