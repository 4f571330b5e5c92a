// amplisolve_amd/csrc/host/aseq.cpp -- .PILEUP.ASEQ ingest: file list, visit order, parallel parse straight into the
// device record layout, chunked hand-over to the pipelines (replaces the parse loops EE:1100-1149 and VC:699-752).
#include <fcntl.h>
#include <glob.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
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

std::vector<std::pair<std::string, std::string>> shard_of_files(const std::vector<std::pair<std::string, std::string>> &files, int shard_index,
                                                                int shard_count, int *first)
{
    // contiguous range of the visit order; earlier shards take the remainder (dist.py::shard_range)
    const int n = (int)files.size(), base = n / shard_count, rem = n % shard_count;
    const int lo = shard_index * base + std::min(shard_index, rem), hi = lo + base + (shard_index < rem ? 1 : 0);
    if (first) *first = lo;
    return std::vector<std::pair<std::string, std::string>>(files.begin() + lo, files.begin() + hi);
}

// ---------------------------------------------------------------------------------------------------------
// record layouts on the host: the packer writes what the kernels read (include/amplisolve_hip.h)
// ---------------------------------------------------------------------------------------------------------
size_t record_bytes(int layout) { return layout == AMPLI_RECORDS_U24 ? 24 : (layout == AMPLI_RECORDS_U16 ? 16 : 32); }

static inline void put_record(int layout, char *dst, const int32_t rec[8])
{
    if (layout == AMPLI_RECORDS_U24) {
        // 8 x 24-bit little-endian fields as three 64-bit words
        const uint64_t f0 = (uint32_t)rec[0], f1 = (uint32_t)rec[1], f2 = (uint32_t)rec[2], f3 = (uint32_t)rec[3], f4 = (uint32_t)rec[4],
                       f5 = (uint32_t)rec[5], f6 = (uint32_t)rec[6], f7 = (uint32_t)rec[7];
        const uint64_t w0 = f0 | (f1 << 24) | (f2 << 48);
        const uint64_t w1 = (f2 >> 16) | (f3 << 8) | (f4 << 32) | (f5 << 56);
        const uint64_t w2 = (f5 >> 8) | (f6 << 16) | (f7 << 40);
        memcpy(dst, &w0, 8); memcpy(dst + 8, &w1, 8); memcpy(dst + 16, &w2, 8);
    } else if (layout == AMPLI_RECORDS_U16) {
        uint16_t v[8];
        for (int j = 0; j < 8; ++j) v[j] = (uint16_t)rec[j];
        memcpy(dst, v, 16);
    } else {
        memcpy(dst, rec, 32);
    }
}

void fill_absent(int layout, char *dst, size_t n_records)
{
    if (!n_records) return;
    int32_t rec[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    rec[0] = layout == AMPLI_RECORDS_U24 ? 0xFFFFFF : (layout == AMPLI_RECORDS_U16 ? 0xFFFF : AMPLI_ABSENT);
    const size_t rb = record_bytes(layout);
    put_record(layout, dst, rec);
    // doubling copy: one pattern record, then memcpy of what is already there
    size_t done = 1;
    while (done < n_records) {
        const size_t k = std::min(done, n_records - done);
        memcpy(dst + done * rb, dst, k * rb);
        done += k;
    }
}

namespace {

struct Extra {
    uint32_t p, k;
    int32_t line;
    int32_t rec[8];
};

struct FileResult {
    std::vector<Extra> extras;
    std::vector<Irregular> irregular; // .sample is filled in by the caller; .record holds the POSITION here
    int64_t n_lines = 0, n_off = 0, n_irregular = 0, n_malformed = 0;
    bool overflows = false; // a count that the layout being packed cannot hold: the chunk is packed again, wider
    int64_t max_count = 0;  // largest strand count of the file (decides how wide)
    std::string error;
    int error_code = 0;
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
    int digits = 0;
    while (p < e && *p >= '0' && *p <= '9') { x = x * 10 + (*p - '0'); ++p; if (++digits > 17) return false; }
    v = neg ? -x : x;
    return true;
}

// One file.  Columns: chr pos dbsnp MAF ref alt A C G T RD Ars Crs Grs Trs (EE:1149, VC:752); the first line
// is the header (EE:1113, VC:721).  fw = X - Xrs (EE:1155-1158).  The first line of a panel position goes straight
// into dst (this sample's row of P records in `layout`, pre-filled as absent by this function); further lines of the
// same position (overlapping amplicons) are returned as extras.  line: optional [P] data-line index of the primaries.
//
// Two tokenisers feed one consumer.  The PLAIN one is the definition: line end by memchr, whitespace = ' ' '\t' '\r', a token = a run of
// anything else, an integer = optional sign + 1..17 digits (whatever follows the digits starts the next token).  The FAST one (round 5)
// reads a line once, left to right, and only accepts what the plain one provably reads the same way: no leading whitespace, every
// separator exactly one '\t', integers unsigned with 1..17 digits directly followed by their separator; at anything else -- a
// space, a sign, "12.5", a short line, a carriage return inside the line, a blank line -- it hands THAT line to the plain tokeniser.
// It needs no bounds checks: it runs only up to the file's last '\n', and every one of its loops stops at a '\n'.
// AMPLISOLVE_PARSER=plain switches it off (tests/test_stream_ingest.py compares the two on hostile files).
static bool fast_tokeniser_enabled() // asked once per file
{
    const char *e = getenv("AMPLISOLVE_PARSER");
    return !(e && strcmp(e, "plain") == 0);
}

void parse_file_text(const Panel &panel, const std::string &path, int layout, char *dst, int32_t *line, FileResult &out)
{
    const int64_t P = panel.P();
    const size_t rb = record_bytes(layout);
    fill_absent(layout, dst, (size_t)P);
    if (line) std::fill_n(line, (size_t)P, -1);
    std::vector<uint32_t> occ((size_t)P, 0); // occurrences so far of each position in THIS file
    const int64_t max_count = layout == AMPLI_RECORDS_U24 ? 0xFFFFFE : (layout == AMPLI_RECORDS_U16 ? 65534 : INT32_MAX);

    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) { out.error = "Cannot open " + path; out.error_code = AMPLI_E_INVALID; return; }
    struct stat st;
    fstat(fd, &st);
    const size_t len = (size_t)st.st_size;
    const char *base = len ? (const char *)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0) : nullptr;
    if (len && base == MAP_FAILED) { close(fd); out.error = "mmap failed for " + path; out.error_code = AMPLI_E_INVALID; return; }
    if (len) madvise((void *)base, len, MADV_SEQUENTIAL);
    const char *cur = base, *end = base + len;
    struct Unmap { const char *b; size_t n; int fd; ~Unmap() { if (b) munmap((void *)b, n); close(fd); } } unmap{base, len, fd};
    // header
    while (cur < end && *cur != '\n') ++cur;
    if (cur < end) ++cur;

    std::string last_chrom;
    int last_cid = -1;
    int64_t prev_p = -2;
    int32_t line_idx = 0;
    // one tokenised data line: chromosome [c0, c0 + clen), coordinate, v = A C G T RD Ars Crs Grs Trs.  false = the file is refused (out.error set)
    auto consume = [&](const char *c0, const size_t clen, const int64_t pos, const int64_t *v) -> bool {
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
            return true;
        }
        prev_p = pi;
        const int64_t A = v[0], C = v[1], G = v[2], T = v[3], RD = v[4];
        const int64_t w[8] = {A - v[5], C - v[6], G - v[7], T - v[8], v[5], v[6], v[7], v[8]};
        int32_t rec[8];
        for (int j = 0; j < 8; ++j) {
            // 0 <= Xrs <= X and everything inside int32: anything else is not a read count (AMPLI_E_RANGE, as
            // include/amplisolve_hip.h promises of the packer)
            if (w[j] < 0 || w[j] > INT32_MAX) {
                out.error = path + ": data line " + std::to_string(line_idx + 1) + ": a strand count is negative (reverse count above the total) or beyond int32";
                out.error_code = AMPLI_E_RANGE;
                return false;
            }
            if (w[j] > out.max_count) out.max_count = w[j];
            if (w[j] > max_count) out.overflows = true;
            rec[j] = (int32_t)w[j];
        }
        const uint32_t k = occ[pi]++;
        if (A + C + G + T != RD) { // EE:1178-1181, VC:762-765: such a line is used with its own RD column (EE:1229, VC:895)
            ++out.n_irregular;
            if (RD < INT32_MIN || RD > INT32_MAX) {
                out.error = path + ": data line " + std::to_string(line_idx + 1) + ": RD beyond int32";
                out.error_code = AMPLI_E_RANGE;
                return false;
            }
            out.irregular.push_back(Irregular{0u, (uint32_t)pi, k, (int32_t)RD});
        }
        if (k == 0) {
            if (!out.overflows) put_record(layout, dst + (size_t)pi * rb, rec);
            if (line) line[(size_t)pi] = line_idx;
        } else {
            Extra e;
            e.p = (uint32_t)pi; e.k = k; e.line = line_idx;
            memcpy(e.rec, rec, sizeof rec);
            out.extras.push_back(e);
        }
        return true;
    };
    // the plain tokeniser on the line [cur, le)
    auto plain_line = [&](const char *lb, const char *le) -> bool {
        const char *p = skip_ws(lb, le);
        if (p >= le) return true; // a blank line is no line at all
        ++out.n_lines;
        const char *c0 = p, *c1 = skip_tok(p, le);
        p = c1;
        int64_t pos, v[9];
        bool ok = parse_int(p, le, pos);
        for (int i = 0; ok && i < 4; ++i) { p = skip_ws(p, le); const char *q = skip_tok(p, le); ok = q > p; p = q; }
        for (int i = 0; ok && i < 9; ++i) ok = parse_int(p, le, v[i]);
        if (!ok) ++out.n_malformed;
        else if (!consume(c0, (size_t)(c1 - c0), pos, v)) return false;
        ++line_idx;
        return true;
    };
    // every line that ends in a '\n' may go through the fast tokeniser; what follows the file's last '\n' (a last line without one) may not
    const char *fast_end = cur;
    if (fast_tokeniser_enabled() && cur < end) {
        const char *last_nl = (const char *)memrchr(cur, '\n', (size_t)(end - cur));
        if (last_nl) fast_end = last_nl + 1;
    }
    auto is_sep = [](const unsigned char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; };
    // unsigned integer of 1..17 digits at p, directly followed by `term`: value in x, p behind the terminator
    auto fast_int = [](const char *&p, const char term, int64_t &x) -> bool {
        const char *s = p;
        uint64_t a = 0;
        unsigned d;
        while ((d = (unsigned)(unsigned char)*p - '0') <= 9u) { a = a * 10 + d; ++p; }
        if (p == s || p - s > 17 || *p != term) return false;
        x = (int64_t)a;
        ++p;
        return true;
    };
    while (cur < fast_end) {
        const char *p = cur;
        bool ok = !is_sep((unsigned char)*p);
        const char *c0 = p;
        size_t clen = 0;
        int64_t pos = 0, v[9];
        if (ok) {
            while (!is_sep((unsigned char)*p)) ++p;
            clen = (size_t)(p - c0);
            ok = *p == '\t';
            ++p;
        }
        ok = ok && fast_int(p, '\t', pos);
        for (int i = 0; ok && i < 4; ++i) {
            const char *t0 = p;
            while (!is_sep((unsigned char)*p)) ++p;
            ok = p > t0 && *p == '\t';
            ++p;
        }
        for (int i = 0; ok && i < 8; ++i) ok = fast_int(p, '\t', v[i]);
        if (ok) { // the last integer: whatever follows its digits belongs to no column
            const char *s0 = p;
            uint64_t a = 0;
            unsigned d;
            while ((d = (unsigned)(unsigned char)*p - '0') <= 9u) { a = a * 10 + d; ++p; }
            ok = p > s0 && p - s0 <= 17;
            v[8] = (int64_t)a;
        }
        if (ok) {
            if (*p != '\n') p = (const char *)memchr(p, '\n', (size_t)(fast_end - p)); // never NULL: fast_end is behind a '\n'
            ++out.n_lines;
            if (!consume(c0, clen, pos, v)) return;
            ++line_idx;
            cur = p + 1;
        } else {
            const char *le = (const char *)memchr(cur, '\n', (size_t)(fast_end - cur));
            if (!plain_line(cur, le)) return;
            cur = le + 1;
        }
    }
    while (cur < end) {
        const char *le = (const char *)memchr(cur, '\n', (size_t)(end - cur));
        if (!le) le = end;
        if (!plain_line(cur, le)) return;
        cur = le < end ? le + 1 : end;
    }
}

// ---------------------------------------------------------------------------------------------------------
// The binary record cache (opt-in: AMPLISOLVE_CACHE=1): `<file>.aseqbin` beside each `.PILEUP.ASEQ` holds what
// parse_file_text produced for it against THIS panel -- the P packed records in the narrowest layout the file's counts fit, the
// data-line index of each, the extra occurrences, the lines with their own RD column and the line statistics -- so that a
// repeated run over the same files (another C value, another coverage cutoff, the calling step after a changed error table)
// skips the text.  A cache file is used only if its header matches: format version, the panel's digest (chromosome names
// and coordinates of its P positions, in order), P, and the text file's size and modification time (ns); anything else --
// also a truncated or unreadable file -- falls back to the text and rewrites it.  Outputs are byte-identical with and without
// (tests/test_stream_ingest.py, tests/test_gpu_cli.py).  Written through a temporary name + rename; an unwritable directory
// just means no cache.  `*.aseqbin` does not match the `*.ASEQ` of the file list (EE:556), the reference's or ours.
// ---------------------------------------------------------------------------------------------------------
struct CacheHeader {
    char magic[8]; // "AMPLBIN1"
    uint32_t layout, n_extras, n_irregular, reserved;
    int64_t P;
    uint64_t panel_digest;
    int64_t text_size, text_mtime_ns;
    int64_t n_lines, n_off, n_irregular_lines, n_malformed, max_count;
};
static_assert(sizeof(CacheHeader) == 96, "cache header layout");

bool cache_enabled()
{
    const char *e = getenv("AMPLISOLVE_CACHE");
    return e && *e && !(e[0] == '0' && e[1] == 0);
}

inline int layout_width(int lay) { return lay == AMPLI_RECORDS_U16 ? 0 : (lay == AMPLI_RECORDS_U24 ? 1 : 2); }
inline int layout_needed(int64_t m) { return m <= 65534 ? AMPLI_RECORDS_U16 : (m <= 0xFFFFFE ? AMPLI_RECORDS_U24 : AMPLI_RECORDS_I32); }

// one record of `layout` -> the eight int32 counts of the interchange layout (absent: rec[0] = AMPLI_ABSENT)
inline void get_record(int layout, const char *src, int32_t rec[8])
{
    if (layout == AMPLI_RECORDS_U16) {
        uint16_t v[8];
        memcpy(v, src, 16);
        for (int j = 0; j < 8; ++j) rec[j] = v[j];
        if (v[0] == 0xFFFF) rec[0] = AMPLI_ABSENT;
    } else if (layout == AMPLI_RECORDS_U24) {
        const unsigned char *b = (const unsigned char *)src;
        for (int j = 0; j < 8; ++j) rec[j] = (int32_t)(b[3 * j] | (b[3 * j + 1] << 8) | (b[3 * j + 2] << 16));
        if (rec[0] == 0xFFFFFF) rec[0] = AMPLI_ABSENT;
    } else {
        memcpy(rec, src, 32);
    }
}

// records of one layout into another that is wide enough for them (absent markers translated)
void convert_records(int from, const char *src, int to, char *dst, size_t n)
{
    if (from == to) { memcpy(dst, src, n * record_bytes(to)); return; }
    const size_t fb = record_bytes(from), tb = record_bytes(to);
    const int32_t absent_to = to == AMPLI_RECORDS_U24 ? 0xFFFFFF : (to == AMPLI_RECORDS_U16 ? 0xFFFF : AMPLI_ABSENT);
    for (size_t i = 0; i < n; ++i) {
        int32_t rec[8];
        get_record(from, src + i * fb, rec);
        if (rec[0] == AMPLI_ABSENT) rec[0] = absent_to;
        put_record(to, dst + i * tb, rec);
    }
}

bool stat_text(const std::string &path, int64_t &size, int64_t &mtime_ns)
{
    struct stat st;
    if (stat(path.c_str(), &st) != 0) return false;
    size = (int64_t)st.st_size;
    mtime_ns = (int64_t)st.st_mtim.tv_sec * 1000000000ll + (int64_t)st.st_mtim.tv_nsec;
    return true;
}

// true: dst / line / out hold what parse_file_text would have produced (out.overflows set when the file needs wider records
// than `layout`, exactly as the text parser reports it)
bool cache_load(const Panel &panel, uint64_t digest, const std::string &path, int layout, char *dst, int32_t *line, FileResult &out)
{
    int64_t tsize, tmtime;
    if (!stat_text(path, tsize, tmtime)) return false;
    const std::string cpath = path + ".aseqbin";
    const int fd = open(cpath.c_str(), O_RDONLY);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(CacheHeader)) { close(fd); return false; }
    const size_t len = (size_t)st.st_size;
    const char *base = (const char *)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (base == MAP_FAILED) return false;
    bool ok = false;
    CacheHeader h;
    memcpy(&h, base, sizeof h);
    const int64_t P = panel.P();
    if (memcmp(h.magic, "AMPLBIN1", 8) == 0 && h.P == P && h.panel_digest == digest && h.text_size == tsize && h.text_mtime_ns == tmtime &&
        (h.layout == AMPLI_RECORDS_U16 || h.layout == AMPLI_RECORDS_U24 || h.layout == AMPLI_RECORDS_I32) && h.reserved == 0) {
        const size_t rb = record_bytes((int)h.layout);
        const size_t need = sizeof h + (size_t)P * rb + (size_t)P * sizeof(int32_t) + (size_t)h.n_extras * sizeof(Extra) + (size_t)h.n_irregular * sizeof(Irregular);
        if (need == len) {
            out.n_lines = h.n_lines; out.n_off = h.n_off; out.n_irregular = h.n_irregular_lines; out.n_malformed = h.n_malformed;
            out.max_count = h.max_count;
            if (layout_width((int)h.layout) > layout_width(layout)) {
                out.overflows = true; // the chunk is packed again, wider (ChunkStream::Impl::fill)
            } else {
                const char *q = base + sizeof h;
                convert_records((int)h.layout, q, layout, dst, (size_t)P);
                q += (size_t)P * rb;
                if (line) memcpy(line, q, (size_t)P * sizeof(int32_t));
                q += (size_t)P * sizeof(int32_t);
                out.extras.resize(h.n_extras);
                if (h.n_extras) memcpy(out.extras.data(), q, (size_t)h.n_extras * sizeof(Extra));
                q += (size_t)h.n_extras * sizeof(Extra);
                out.irregular.resize(h.n_irregular);
                if (h.n_irregular) memcpy(out.irregular.data(), q, (size_t)h.n_irregular * sizeof(Irregular));
            }
            ok = true;
        }
    }
    munmap((void *)base, len);
    return ok;
}

void cache_store(const Panel &panel, uint64_t digest, const std::string &path, int layout, const char *recs, const int32_t *line, const FileResult &r)
{
    int64_t tsize, tmtime;
    if (!stat_text(path, tsize, tmtime)) return;
    const int64_t P = panel.P();
    CacheHeader h;
    memset(&h, 0, sizeof h);
    memcpy(h.magic, "AMPLBIN1", 8);
    h.layout = (uint32_t)layout_needed(r.max_count); // the narrowest layout the file fits, whatever the chunk was packed in
    h.n_extras = (uint32_t)r.extras.size(); h.n_irregular = (uint32_t)r.irregular.size();
    h.P = P; h.panel_digest = digest; h.text_size = tsize; h.text_mtime_ns = tmtime;
    h.n_lines = r.n_lines; h.n_off = r.n_off; h.n_irregular_lines = r.n_irregular; h.n_malformed = r.n_malformed; h.max_count = r.max_count;
    std::vector<char> narrow;
    const char *body = recs;
    if ((int)h.layout != layout) {
        narrow.resize((size_t)P * record_bytes((int)h.layout));
        convert_records(layout, recs, (int)h.layout, narrow.data(), (size_t)P);
        body = narrow.data();
    }
    char tag[64];
    snprintf(tag, sizeof tag, ".tmp%ld_%lx", (long)getpid(), (unsigned long)std::hash<std::thread::id>()(std::this_thread::get_id()));
    const std::string cpath = path + ".aseqbin", tmp = cpath + tag;
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return; // read-only data directory: no cache
    bool ok = fwrite(&h, sizeof h, 1, f) == 1 && fwrite(body, record_bytes((int)h.layout), (size_t)P, f) == (size_t)P &&
              fwrite(line, sizeof(int32_t), (size_t)P, f) == (size_t)P;
    if (ok && h.n_extras) ok = fwrite(r.extras.data(), sizeof(Extra), h.n_extras, f) == h.n_extras;
    if (ok && h.n_irregular) ok = fwrite(r.irregular.data(), sizeof(Irregular), h.n_irregular, f) == h.n_irregular;
    ok = (fclose(f) == 0) && ok;
    if (!ok || rename(tmp.c_str(), cpath.c_str()) != 0) unlink(tmp.c_str());
}

// One file through the cache when it is on and valid, else through the text parser (which then refreshes the cache)
void parse_file(const Panel &panel, uint64_t digest, const std::string &path, int layout, char *dst, int32_t *line, FileResult &out)
{
    if (!cache_enabled()) { parse_file_text(panel, path, layout, dst, line, out); return; }
    if (cache_load(panel, digest, path, layout, dst, line, out)) return;
    out = FileResult();
    std::vector<int32_t> own_line;
    int32_t *ln = line;
    if (!ln) { own_line.resize((size_t)panel.P()); ln = own_line.data(); } // the cache always carries the line indices (the calling step needs them)
    parse_file_text(panel, path, layout, dst, ln, out);
    if (out.error.empty() && !out.overflows) cache_store(panel, digest, path, layout, dst, ln, out);
}

// chromosome names + coordinates of the panel's positions, in order (FNV-1a): what a cache file was made against
uint64_t panel_digest(const Panel &panel)
{
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    };
    for (const std::string &c : panel.chroms) { mix(c.data(), c.size()); mix("\0", 1); }
    const int64_t P = panel.P();
    mix(&P, sizeof P);
    if (P) { mix(panel.pos_chrom.data(), (size_t)P * sizeof(int32_t)); mix(panel.pos_coord.data(), (size_t)P * sizeof(int32_t)); }
    return h;
}

void *host_alloc(size_t bytes, bool &pinned, bool critical = true)
{
    PhaseClock::Scope sc("pinned_alloc", critical);
    const HipApi *api = hip_api();
    void *mem = nullptr;
    if (api && api->device_count() > 0 && api->pinned_alloc(bytes ? bytes : 32, &mem) == AMPLI_OK) { pinned = true; return mem; }
    pinned = false;
    return aligned_alloc(256, (bytes + 255) / 256 * 256 + 256);
}

void host_free(void *p, bool pinned)
{
    if (!p) return;
    const HipApi *api = pinned ? hip_api() : nullptr;
    if (pinned && api) api->pinned_free(p);
    else free(p);
}

// the ring's buffers: plain page-aligned memory (2 MiB aligned so the kernel may back it with huge pages); no HIP call
constexpr size_t kBufAlign = 2u << 20;
void *ring_alloc(size_t &bytes)
{
    bytes = (std::max<size_t>(bytes, 1) + kBufAlign - 1) / kBufAlign * kBufAlign;
    void *p = aligned_alloc(kBufAlign, bytes);
    if (p) madvise(p, bytes, MADV_HUGEPAGE);
    return p;
}

void ring_free(void *&p, bool &pinned)
{
    if (!p) return;
    if (pinned) {
        PhaseClock::Scope sc("host_unregister", false);
        if (const HipApi *api = hip_api()) api->host_unregister(p);
    }
    free(p);
    p = nullptr;
    pinned = false;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------
// ChunkStream: the cohort as a sequence of chunks of consecutive samples, parsed by a producer thread (+ workers)
// into a small ring of pinned buffers while the consumer uploads / reduces the previous chunk.
// ---------------------------------------------------------------------------------------------------------
Chunk::~Chunk()
{
    ring_free(prim, prim_pinned);
    ring_free(ext, ext_pinned);
}

void Chunk::pin(ampli_ctx *ctx)
{
    const HipApi *api = hip_api();
    if (!api) return;
    PhaseClock::Scope sc("host_register");
    bool refused = false;
    if (prim && !prim_pinned) { if (api->host_register(ctx, prim, prim_cap) == AMPLI_OK) prim_pinned = true; else refused = true; }
    if (ext && !ext_pinned) { if (api->host_register(ctx, ext, ext_cap) == AMPLI_OK) ext_pinned = true; else refused = true; }
    if (refused) { // the upload still works (the runtime stages it), only slower: say so once instead of degrading in silence
        static std::once_flag once;
        std::call_once(once, [&] { std::cerr << "note: a record buffer of " << (prim_cap >> 20) << " MB could not be pinned (hipHostRegister refused: locked-memory limit?); "
                                                "its uploads are staged by the runtime" << std::endl; });
    }
}

struct ChunkStream::Impl {
    const Panel &panel;
    std::vector<std::pair<std::string, std::string>> files;
    int n_threads = 1;
    bool keep_line = false;
    int per_chunk = 1, n_chunks = 0;
    uint64_t digest = 0;                  // of the panel's positions: what a `.aseqbin` cache file must have been made against
    int floor_layout = AMPLI_RECORDS_U16; // narrowest layout the packer may use (AMPLISOLVE_RECORDS)
    int start_layout = AMPLI_RECORDS_U16; // layout the next chunk is packed in first
    std::vector<std::unique_ptr<Chunk>> slots;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<int> free_slots, ready; // ready: slot indices in chunk order
    size_t ring_budget = 0;             // bytes the slots in circulation may hold: the ring as it was sized (in the narrowest layout allowed)
    int outstanding = 0;                // chunks handed to the consumer and not released yet
    bool done = false, stop = false, failed = false;
    Error err{0, ""};
    std::thread producer;
    double parse_seconds = 0;

    explicit Impl(const Panel &p) : panel(p) {}

    void fill(Chunk &c, int ci)
    {
        const int64_t P = panel.P();
        const int lo = ci * per_chunk, hi = std::min((int)files.size(), lo + per_chunk), n = hi - lo;
        c.index = ci; c.first = lo; c.n = n; c.P = P; c.last = ci == n_chunks - 1;
        c.n_lines = c.n_offpanel = c.n_irregular = c.n_malformed = 0;
        c.irregular.clear();
        // The narrowest layout the chunk's counts fit (identical results in all three; the kernels' time follows the bytes):
        // 8 x uint16 = 16 B up to 65534 reads per strand and base, 8 x 24 bit = 24 B up to 2^24 - 2, else int32.  A chunk is
        // packed optimistically in the layout the previous one needed and packed again, wider, if a count does not fit --
        // at most once, because the first pass has seen the largest count.
        int layout = start_layout;
        int64_t mx = 0; // largest strand count of the chunk
        auto width = [](int lay) { return lay == AMPLI_RECORDS_U16 ? 0 : (lay == AMPLI_RECORDS_U24 ? 1 : 2); };
        auto needed = [](int64_t m) { return m <= 65534 ? AMPLI_RECORDS_U16 : (m <= 0xFFFFFE ? AMPLI_RECORDS_U24 : AMPLI_RECORDS_I32); };
        std::vector<FileResult> res;
        for (int attempt = 0; attempt < 2; ++attempt) {
            res.assign((size_t)n, FileResult());
            const size_t rb = record_bytes(layout);
            if ((size_t)n * (size_t)P * rb > c.prim_cap) { // the buffers are sized for the narrowest layout; a wider chunk grows its own
                ring_free(c.prim, c.prim_pinned);
                c.prim_cap = (size_t)per_chunk * (size_t)P * rb;
                c.prim = ring_alloc(c.prim_cap);
                if (!c.prim) throw Error{AMPLI_E_NOMEM, "cannot allocate the record buffers"};
            }
            if (keep_line) c.line_prim.assign((size_t)n * P, -1);
            std::atomic<int> next{0};
            auto work = [&] {
                for (int i; (i = next.fetch_add(1)) < n;)
                    parse_file(panel, digest, files[(size_t)lo + i].first, layout, (char *)c.prim + (size_t)i * P * rb,
                               keep_line ? c.line_prim.data() + (size_t)i * P : nullptr, res[(size_t)i]);
            };
            const int nt = std::max(1, std::min(n_threads, n));
            std::vector<std::thread> th;
            for (int t = 1; t < nt; ++t) th.emplace_back(work);
            work();
            for (auto &t : th) t.join();
            bool widen = false;
            mx = 0;
            for (auto &r : res) {
                if (!r.error.empty()) throw Error{r.error_code ? r.error_code : AMPLI_E_INVALID, r.error};
                widen |= r.overflows;
                mx = std::max(mx, r.max_count);
            }
            if (!widen) break;
            if (layout == AMPLI_RECORDS_I32) throw Error{AMPLI_E_RANGE, "count beyond int32"};
            layout = needed(mx);
        }
        c.layout = layout;
        // the next chunk starts in what THIS one needed (never narrower than AMPLISOLVE_RECORDS allows): a deep cohort is not
        // parsed twice per chunk, and one outlier widens its own chunk and at most the next
        start_layout = width(needed(mx)) < width(floor_layout) ? floor_layout : needed(mx);
        // extras of this chunk: slots for the largest multiplicity any of ITS files shows at a position
        std::vector<uint32_t> mult((size_t)P, 0);
        bool any_extra = false;
        for (auto &r : res)
            for (auto &e : r.extras) { any_extra = true; if (e.k > mult[e.p]) mult[e.p] = e.k; }
        c.dup_off.assign((size_t)P + 1, 0);
        if (any_extra)
            for (int64_t p = 0; p < P; ++p) c.dup_off[p + 1] = c.dup_off[p] + mult[p];
        c.E = c.dup_off[P];
        c.ext_pos.assign((size_t)c.E, 0);
        for (int64_t p = 0; p < P && any_extra; ++p)
            for (uint32_t e = c.dup_off[p]; e < c.dup_off[p + 1]; ++e) c.ext_pos[e] = (uint32_t)p;
        const size_t rb = record_bytes(layout);
        const size_t need = (size_t)n * (size_t)c.E * rb;
        if (need > c.ext_cap) {
            ring_free(c.ext, c.ext_pinned);
            c.ext_cap = need + need / 2 + 4096;
            c.ext = ring_alloc(c.ext_cap);
            if (!c.ext) throw Error{AMPLI_E_NOMEM, "cannot allocate the extra-occurrence records"};
        }
        if (c.E) fill_absent(layout, (char *)c.ext, (size_t)n * (size_t)c.E);
        if (keep_line) c.line_ext.assign((size_t)n * c.E, -1);
        for (int i = 0; i < n; ++i) {
            FileResult &r = res[(size_t)i];
            for (auto &e : r.extras) {
                const size_t slot = (size_t)c.dup_off[e.p] + (e.k - 1);
                put_record(layout, (char *)c.ext + ((size_t)i * c.E + slot) * rb, e.rec);
                if (keep_line) c.line_ext[(size_t)i * c.E + slot] = e.line;
            }
            for (auto &ir : r.irregular) {
                Irregular x = ir;
                x.sample = (uint32_t)i;
                // record slot of occurrence k of position p inside this chunk's layout
                x.record = ir.occurrence == 0 ? ir.record : (uint32_t)(P + c.dup_off[ir.record] + (ir.occurrence - 1));
                c.irregular.push_back(x);
            }
            c.n_lines += r.n_lines; c.n_offpanel += r.n_off; c.n_irregular += r.n_irregular; c.n_malformed += r.n_malformed;
        }
    }

    void run()
    {
        try {
            for (int ci = 0; ci < n_chunks; ++ci) {
                int slot;
                {
                    std::unique_lock<std::mutex> lk(mu);
                    // The ring was sized in the narrowest layout the packer may use.  Once the cohort has turned out to need wider records
                    // (24 or 32 bytes each) every slot that is refilled grows by half or doubles -- and is pinned: with the default 16 slots of
                    // 128 MB that was 3-4 GiB of locked memory for an int32 cohort.  So the number of slots in circulation shrinks with the
                    // layout in use (at least four stay: what the pipeline needs to keep the parsers ahead of the uploads).
                    auto may_take = [&] {
                        if (free_slots.empty()) return false;
                        const size_t slot_bytes = (size_t)per_chunk * (size_t)panel.P() * record_bytes(start_layout);
                        const size_t max_live = std::max<size_t>(4, ring_budget / std::max<size_t>(1, slot_bytes));
                        return slots.size() - free_slots.size() < max_live;
                    };
                    cv.wait(lk, [&] { return stop || may_take(); });
                    if (stop) return;
                    slot = free_slots.back();
                    free_slots.pop_back();
                }
                const auto t0 = std::chrono::steady_clock::now();
                fill(*slots[(size_t)slot], ci);
                const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                {
                    std::lock_guard<std::mutex> lk(mu);
                    parse_seconds += dt;
                    ready.push_back(slot);
                }
                cv.notify_all();
            }
        } catch (const Error &e) {
            std::lock_guard<std::mutex> lk(mu);
            err = e;
            failed = true;
        } catch (const std::exception &e) {
            std::lock_guard<std::mutex> lk(mu);
            err = Error{AMPLI_E_NOMEM, std::string("ingest: ") + e.what()};
            failed = true;
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            done = true;
        }
        cv.notify_all();
    }
};

ChunkStream::ChunkStream(const Panel &panel, std::vector<std::pair<std::string, std::string>> files, int n_threads, bool keep_line_no,
                         size_t chunk_bytes, int n_slots)
    : im(new Impl(panel))
{
    im->files = std::move(files);
    if (cache_enabled()) im->digest = panel_digest(panel);
    // default: the cores this process may run on, at most 16 (one GPU's share of a node; AMPLISOLVE_THREADS overrides)
    if (n_threads <= 0) n_threads = std::min(16, (int)std::thread::hardware_concurrency());
    im->n_threads = std::max(1, n_threads);
    im->keep_line = keep_line_no;
    if (const char *e = getenv("AMPLISOLVE_RECORDS")) { // narrowest record layout the packer may use: u16 (default) | u24 | i32
        const std::string v(e);
        if (v == "u24") im->floor_layout = im->start_layout = AMPLI_RECORDS_U24;
        else if (v == "i32") im->floor_layout = im->start_layout = AMPLI_RECORDS_I32;
        else if (v != "u16" && v != "auto" && !v.empty()) throw Error{AMPLI_E_INVALID, "AMPLISOLVE_RECORDS must be u16, u24 or i32"};
    }
    const int64_t P = panel.P();
    const int S = (int)im->files.size();
    // samples per chunk: about chunk_bytes of records IN THE NARROWEST LAYOUT ALLOWED, at least one sample, and not so few
    // that the parser threads idle.  A chunk whose counts need wider records keeps its sample count, so AMPLISOLVE_CHUNK_MB = 128
    // means up to 192 MB (24-bit fields) or 256 MB (int32) for such a chunk, in each of the ring's slots and of the device's;
    // the slot is re-allocated (plain memory; pinned again on its next upload) by the producer, off the consumer's path
    int per = (int)std::max<int64_t>(1, (int64_t)(chunk_bytes / (size_t)std::max<int64_t>(1, P * (int64_t)record_bytes(im->start_layout))));
    per = std::max(per, std::min(S, im->n_threads));
    per = std::min(per, std::max(1, S));
    im->per_chunk = per;
    im->n_chunks = S == 0 ? 0 : (S + per - 1) / per;
    n_slots = std::max(1, std::min(n_slots, std::max(1, im->n_chunks)));
    for (int i = 0; i < n_slots; ++i) {
        std::unique_ptr<Chunk> c(new Chunk());
        c->slot = i;
        c->prim_cap = (size_t)per * (size_t)P * record_bytes(im->start_layout); // a chunk that needs wider records grows its buffer (fill)
        c->prim = ring_alloc(c->prim_cap);
        if (!c->prim) throw Error{AMPLI_E_NOMEM, "cannot allocate the record buffers"};
        im->ring_budget += c->prim_cap;
        im->slots.push_back(std::move(c));
        im->free_slots.push_back(i);
    }
    im->producer = std::thread([this] { im->run(); });
}

void ChunkStream::shutdown()
{
    {
        std::lock_guard<std::mutex> lk(im->mu);
        im->stop = true;
    }
    im->cv.notify_all();
    if (im->producer.joinable()) im->producer.join();
    im->slots.clear();
}

void ChunkStream::abandon()
{
    {
        std::lock_guard<std::mutex> lk(im->mu);
        im->stop = true;
    }
    im->cv.notify_all();
    if (im->producer.joinable()) im->producer.join();
    for (auto &c : im->slots) { // the buffers stay mapped (and pinned) until the process ends
        if (c) { c->prim = nullptr; c->ext = nullptr; c->prim_pinned = c->ext_pinned = false; }
    }
    im->slots.clear();
}

ChunkStream::~ChunkStream()
{
    shutdown();
    delete im;
}

int ChunkStream::samples_per_chunk() const { return im->per_chunk; }
int ChunkStream::chunks() const { return im->n_chunks; }
double ChunkStream::parse_seconds() const
{
    std::lock_guard<std::mutex> lk(im->mu);
    return im->parse_seconds;
}

Chunk *ChunkStream::next()
{
    std::unique_lock<std::mutex> lk(im->mu);
    im->cv.wait(lk, [&] { return !im->ready.empty() || im->done; });
    if (im->ready.empty()) {
        if (im->failed) throw im->err;
        return nullptr;
    }
    // The consumers map ring slot k to device buffer k mod kDevSlots (pipeline.cpp) and that is only safe because a chunk is uploaded,
    // consumed and waited for before the next one is taken: checked here rather than assumed.
    if (im->outstanding != 0) throw Error{AMPLI_E_INVALID, "ChunkStream::next(): the previous chunk was not released (its device buffers would be overwritten)"};
    const int slot = im->ready.front();
    im->ready.erase(im->ready.begin());
    im->outstanding = 1;
    return im->slots[(size_t)slot].get();
}

void ChunkStream::release(Chunk *c)
{
    {
        std::lock_guard<std::mutex> lk(im->mu);
        im->free_slots.push_back(c->slot);
        im->outstanding = 0;
    }
    im->cv.notify_all();
}

// ---------------------------------------------------------------------------------------------------------
// cohort_load: the whole directory as ONE dense int32 array [S][P+E][8] (the interchange layout the oracle, the tests
// and the Python harness speak).  The command lines stream chunks instead (ChunkStream above).
// ---------------------------------------------------------------------------------------------------------
void cohort_load(const Panel &panel, const std::string &dir, const std::string &list_file, int n_threads, bool keep_line_no,
                 bool print_irregular, Cohort &out, int shard_index, int shard_count)
{
    auto files = list_count_files(dir, list_file);
    out.total_samples = (int)files.size();
    out.first_sample = 0;
    if (shard_count > 1) files = shard_of_files(files, shard_index, shard_count, &out.first_sample);
    const int S = (int)files.size();
    const int64_t P = panel.P();
    out.paths.clear(); out.names.clear();
    for (auto &f : files) { out.paths.push_back(f.first); out.names.push_back(f.second); }
    out.P = P;

    const uint64_t digest = cache_enabled() ? panel_digest(panel) : 0;
    std::vector<FileResult> res((size_t)S);
    std::vector<std::vector<int32_t>> prim((size_t)S), lines((size_t)S);
    if (n_threads <= 0) n_threads = (int)std::thread::hardware_concurrency();
    if (n_threads > S) n_threads = S;
    if (n_threads < 1) n_threads = 1;
    std::atomic<int> next{0};
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t)
        th.emplace_back([&] {
            for (int s; (s = next.fetch_add(1)) < S;) {
                prim[(size_t)s].resize((size_t)P * 8);
                if (keep_line_no) lines[(size_t)s].resize((size_t)P);
                parse_file(panel, digest, out.paths[(size_t)s], AMPLI_RECORDS_I32, (char *)prim[(size_t)s].data(),
                           keep_line_no ? lines[(size_t)s].data() : nullptr, res[(size_t)s]);
            }
        });
    for (auto &t : th) t.join();
    for (int s = 0; s < S; ++s)
        if (!res[s].error.empty()) throw Error{res[s].error_code ? res[s].error_code : AMPLI_E_INVALID, res[s].error};
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
    out.recs = (int32_t *)host_alloc(bytes, out.pinned);
    if (!out.recs) throw Error{AMPLI_E_NOMEM, "cannot allocate the record array"};
    if (keep_line_no) out.line_no.assign((size_t)S * R, -1);
    out.irregular.clear();
    for (int s = 0; s < S; ++s) {
        int32_t *dst = out.recs + (size_t)s * R * 8;
        memcpy(dst, prim[(size_t)s].data(), (size_t)P * 8 * sizeof(int32_t));
        fill_absent(AMPLI_RECORDS_I32, (char *)(dst + (size_t)P * 8), (size_t)out.E);
        if (keep_line_no) memcpy(&out.line_no[(size_t)s * R], lines[(size_t)s].data(), (size_t)P * sizeof(int32_t));
        for (auto &e : res[s].extras) {
            const size_t slot = (size_t)P + out.dup_off[e.p] + (e.k - 1);
            memcpy(dst + slot * 8, e.rec, sizeof e.rec);
            if (keep_line_no) out.line_no[(size_t)s * R + slot] = e.line;
        }
        for (auto &ir : res[s].irregular) {
            Irregular x = ir;
            x.sample = (uint32_t)s;
            x.record = ir.occurrence == 0 ? ir.record : (uint32_t)(P + out.dup_off[ir.record] + (ir.occurrence - 1));
            out.irregular.push_back(x);
        }
        out.n_lines += res[s].n_lines; out.n_offpanel += res[s].n_off;
        out.n_irregular += res[s].n_irregular; out.n_malformed += res[s].n_malformed;
        std::vector<int32_t>().swap(prim[(size_t)s]); // release as we go
    }
}

} // namespace ampli
