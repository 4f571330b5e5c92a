// amplisolve_amd/csrc/host/bam.cpp -- computeCounts: BAM alignments -> <sample>.PILEUP.ASEQ, the step upstream of the two
// AmpliSolve programs (/root/reference/Execution_examples.md:16-46; the reference ships it as a source-less binary, a
// "simplified version of ASEQ" in PILEUP mode: vcf= bam= threads= mbq= mrq= mdc= out=).
//   host: BGZF container (RFC 1952 members with the 'BC' extra field, inflated with zlib on `threads` threads straight into
//         pinned memory), BAM header, record boundaries + sanity checks, the panel (positions of the VCF-like file), the
//         ASEQ writer;
//   device (ampli_pileup_count): decoding of the alignment records and the per-position base x strand counting.
// Parity: UNPINNED -- the upstream program cannot run here (Mach-O, no source) and the reference holds no BAM fixture; the
// semantics restate the published PILEUP mode (htslib pileup defaults) and are checked against an independent Python
// restatement (oracle/pileup_oracle.py) on synthetic BAM files.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <thread>

#include "hip_loader.hpp"
#include "host.hpp"

namespace ampli {

namespace {

struct Mapped { // read-only mapping of a whole file
    const unsigned char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    explicit Mapped(const std::string &path)
    {
        fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw Error{AMPLI_E_INVALID, "cannot open " + path};
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); throw Error{AMPLI_E_INVALID, "cannot stat " + path}; }
        n = (size_t)st.st_size;
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { close(fd); throw Error{AMPLI_E_INVALID, "cannot map " + path}; }
            p = (const unsigned char *)m;
        }
    }
    ~Mapped()
    {
        if (p) munmap((void *)p, n);
        if (fd >= 0) close(fd);
    }
    Mapped(const Mapped &) = delete;
    Mapped &operator=(const Mapped &) = delete;
};

inline uint32_t le32(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }
inline uint32_t le16(const unsigned char *p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8); }

struct Block {
    size_t cdata;      // offset of the deflate stream in the file
    uint32_t clen;     // its length
    uint32_t isize;    // uncompressed length
};

// walk the BGZF members of the file (SAM/BAM specification, section 4.1)
std::vector<Block> bgzf_blocks(const Mapped &f, const std::string &path)
{
    std::vector<Block> out;
    size_t o = 0;
    while (o < f.n) {
        if (f.n - o < 18 || f.p[o] != 0x1f || f.p[o + 1] != 0x8b || f.p[o + 2] != 8 || !(f.p[o + 3] & 4))
            throw Error{AMPLI_E_INVALID, path + ": not a BGZF file (bad member header at byte " + std::to_string(o) + ")"};
        const uint32_t xlen = le16(f.p + o + 10);
        if (f.n - o < 12 + (size_t)xlen + 8) throw Error{AMPLI_E_INVALID, path + ": truncated BGZF member"};
        uint32_t bsize = 0;
        bool found = false;
        for (size_t x = o + 12; x + 4 <= o + 12 + xlen;) {
            const uint32_t slen = le16(f.p + x + 2);
            if (f.p[x] == 'B' && f.p[x + 1] == 'C' && slen == 2 && x + 6 <= o + 12 + xlen) { bsize = le16(f.p + x + 4); found = true; }
            x += 4 + slen;
        }
        const size_t total = (size_t)bsize + 1;
        if (!found || total < 12 + (size_t)xlen + 8 || f.n - o < total) throw Error{AMPLI_E_INVALID, path + ": bad BGZF block size"};
        Block b;
        b.cdata = o + 12 + xlen;
        b.clen = (uint32_t)(total - 12 - xlen - 8);
        b.isize = le32(f.p + o + total - 4);
        if (b.isize > 65536) throw Error{AMPLI_E_INVALID, path + ": BGZF block larger than 64 KiB"};
        if (b.isize) out.push_back(b); // the empty EOF marker block carries nothing
        o += total;
    }
    return out;
}

void inflate_block(const unsigned char *src, uint32_t clen, unsigned char *dst, uint32_t isize, const std::string &path)
{
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit2(&zs, -15) != Z_OK) throw Error{AMPLI_E_NOMEM, "zlib inflateInit2 failed"};
    zs.next_in = const_cast<unsigned char *>(src);
    zs.avail_in = clen;
    zs.next_out = dst;
    zs.avail_out = isize;
    const int rc = inflate(&zs, Z_FINISH);
    const bool ok = rc == Z_STREAM_END && zs.total_out == isize;
    inflateEnd(&zs);
    if (!ok) throw Error{AMPLI_E_INVALID, path + ": corrupt BGZF block (inflate)"};
}

// blocks [b0, b1) -> dst (contiguous), on n_threads threads
void inflate_range(const Mapped &f, const std::vector<Block> &blocks, size_t b0, size_t b1, unsigned char *dst, int n_threads, const std::string &path)
{
    std::vector<size_t> off(b1 - b0 + 1, 0);
    for (size_t i = b0; i < b1; ++i) off[i - b0 + 1] = off[i - b0] + blocks[i].isize;
    std::atomic<size_t> next{b0};
    std::atomic<bool> failed{false};
    std::string why;
    auto work = [&]() {
        try {
            for (size_t i; (i = next.fetch_add(1)) < b1 && !failed.load();)
                inflate_block(f.p + blocks[i].cdata, blocks[i].clen, dst + off[i - b0], blocks[i].isize, path);
        } catch (const Error &e) {
            if (!failed.exchange(true)) why = e.msg;
        }
    };
    const int nt = std::max(1, std::min<int>(n_threads, (int)(b1 - b0)));
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto &t : th) t.join();
    if (failed.load()) throw Error{AMPLI_E_INVALID, why};
}

struct VcfLine {
    std::string chrom, id, ref, alt;
    int64_t pos = 0;
    int64_t key_index = -1; // index into the sorted unique keys, -1: the chromosome is not in the BAM header
};

std::vector<VcfLine> read_positions(const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw Error{AMPLI_E_INVALID, "cannot open " + path};
    std::vector<VcfLine> out;
    std::string line;
    while (std::getline(in, line)) {
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        VcfLine v;
        std::string pos;
        if (!(ss >> v.chrom >> pos)) continue;
        char *end = nullptr;
        v.pos = strtoll(pos.c_str(), &end, 10);
        if (end == pos.c_str() || v.pos <= 0) continue; // a header line without '#'
        if (!(ss >> v.id)) v.id = ".";
        if (!(ss >> v.ref)) v.ref = ".";
        if (!(ss >> v.alt)) v.alt = ".";
        out.push_back(v);
    }
    return out;
}

struct Header {
    std::vector<std::string> ref_names;
    size_t end = 0; // offset of the first alignment record in the uncompressed stream
};

// BAM header at the start of buf (n bytes available); false: need more bytes
bool parse_header(const unsigned char *buf, size_t n, Header &h, const std::string &path)
{
    if (n < 12) return false;
    if (memcmp(buf, "BAM\1", 4) != 0) throw Error{AMPLI_E_INVALID, path + ": not a BAM file (magic)"};
    const size_t l_text = le32(buf + 4);
    size_t o = 8 + l_text;
    if (n < o + 4) return false;
    const uint32_t n_ref = le32(buf + o);
    o += 4;
    h.ref_names.clear();
    for (uint32_t i = 0; i < n_ref; ++i) {
        if (n < o + 4) return false;
        const size_t l_name = le32(buf + o);
        if (n < o + 4 + l_name + 4) return false;
        h.ref_names.emplace_back((const char *)buf + o + 4, l_name ? l_name - 1 : 0);
        o += 4 + l_name + 4;
    }
    h.end = o;
    return true;
}

// A block_size beyond this is not a record but a corrupt or desynchronised stream (the longest reads there are -- megabases
// of nanopore sequence -- stay two orders of magnitude below).  Treating it as "incomplete, wait for more bytes" would carry
// the rest of the file from batch to batch and end in a silently truncated pileup.
constexpr size_t BAM_MAX_RECORD = (size_t)1 << 28;

// list the complete, well-formed alignment records of buf[from, n); returns the offset of the first incomplete one.
// stream_off: offset of buf[0] in the uncompressed stream (for messages).
size_t scan_records(const unsigned char *buf, size_t from, size_t n, std::vector<uint64_t> &off, int64_t &malformed, const std::string &path,
                    uint64_t stream_off)
{
    size_t o = from;
    while (n - o >= 4) {
        const size_t bs = le32(buf + o);
        if (bs > BAM_MAX_RECORD)
            throw Error{AMPLI_E_INVALID, path + ": corrupt BAM record at uncompressed byte " + std::to_string(stream_off + o) + " (block_size " + std::to_string(bs) + ")"};
        if (n - o - 4 < bs) break; // incomplete: carried over to the next batch
        bool ok = bs >= 32;
        if (ok) {
            const unsigned char *r = buf + o + 4;
            const size_t l_name = r[8], n_cigar = le16(r + 12), l_seq = le32(r + 16);
            ok = (int32_t)l_seq >= 0 && 32 + l_name + 4 * n_cigar + (l_seq + 1) / 2 + l_seq <= bs;
            if (ok && n_cigar) { // query-consuming operations must add up to l_seq (the kernel indexes seq / qual by them)
                size_t q = 0;
                const unsigned char *c = r + 32 + l_name;
                for (size_t i = 0; i < n_cigar; ++i) {
                    const uint32_t v = le32(c + 4 * i), op = v & 15u;
                    if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) q += v >> 4;
                }
                ok = q == l_seq;
            }
        }
        if (ok) off.push_back((uint64_t)o);
        else ++malformed;
        o += 4 + bs;
    }
    return o;
}

struct PinnedBuf {
    const HipApi *api = nullptr;
    unsigned char *p = nullptr;
    size_t cap = 0;
    ~PinnedBuf() { if (p) api->pinned_free(p); }
    void ensure(const HipApi *a, size_t bytes)
    {
        api = a;
        if (bytes <= cap) return;
        if (p) api->pinned_free(p);
        p = nullptr;
        void *q = nullptr;
        if (api->pinned_alloc(bytes, &q) != AMPLI_OK) throw Error{AMPLI_E_NOMEM, "cannot allocate pinned memory for the BAM batch"};
        p = (unsigned char *)q;
        cap = bytes;
    }
};

} // namespace

void bam_scan(const std::string &bam, int n_threads, int64_t stats[4])
{
    Mapped f(bam);
    const auto blocks = bgzf_blocks(f, bam);
    size_t total = 0;
    for (auto &b : blocks) total += b.isize;
    std::vector<unsigned char> buf(total);
    inflate_range(f, blocks, 0, blocks.size(), buf.data(), n_threads > 0 ? n_threads : 4, bam);
    Header h;
    if (!parse_header(buf.data(), buf.size(), h, bam)) throw Error{AMPLI_E_INVALID, bam + ": truncated BAM header"};
    std::vector<uint64_t> off;
    int64_t malformed = 0;
    const size_t end = scan_records(buf.data(), h.end, buf.size(), off, malformed, bam, 0);
    if (end != buf.size())
        throw Error{AMPLI_E_INVALID, bam + ": " + std::to_string(buf.size() - end) + " bytes behind the last complete alignment record (truncated or corrupt file)"};
    stats[0] = (int64_t)off.size();
    stats[1] = (int64_t)total;
    stats[2] = (int64_t)h.ref_names.size();
    stats[3] = malformed;
}

int run_compute_counts(const CcArgs &a)
{
    try {
        if (a.vcf.empty() || a.bam.empty() || a.out_dir.empty()) throw Error{AMPLI_E_INVALID, "vcf=, bam= and out= are required"};
        const int threads = a.threads > 0 ? a.threads : 4;
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_start = now();
        double t_inflate = 0, t_wait = 0;
        std::string why;
        const HipApi *api = hip_api(&why);
        if (!api) throw Error{AMPLI_E_HIP, "libamplisolve_hip.so could not be loaded (" + why + "); there is no CPU fallback"};
        if (api->device_count() <= 0) throw Error{AMPLI_E_HIP, "no MI355X visible; there is no CPU fallback"};
        std::vector<VcfLine> lines = read_positions(a.vcf);
        if (lines.empty()) throw Error{AMPLI_E_INVALID, "no position could be read from " + a.vcf};

        Mapped f(a.bam);
        const auto blocks = bgzf_blocks(f, a.bam);
        if (blocks.empty()) throw Error{AMPLI_E_INVALID, a.bam + ": empty BAM file"};

        struct Ctx {
            const HipApi *api;
            ampli_ctx *ctx = nullptr;
            std::vector<void *> allocs;
            ~Ctx()
            {
                if (ctx) {
                    for (void *p : allocs) api->dev_free(ctx, p);
                    api->ctx_destroy(ctx);
                }
            }
            void check(int rc, const char *what)
            {
                if (rc != AMPLI_OK) throw Error{rc, std::string(what) + ": " + api->strerror_(rc) + (ctx ? std::string(" -- ") + api->last_error(ctx) : "")};
            }
            void *alloc(size_t n)
            {
                void *p = nullptr;
                check(api->dev_alloc(ctx, n ? n : 16, &p), "ampli_dev_alloc");
                allocs.push_back(p);
                return p;
            }
            void release(void *p) // a slot's buffer is replaced by a larger one
            {
                if (!p) return;
                allocs.erase(std::remove(allocs.begin(), allocs.end(), p), allocs.end());
                check(api->dev_free(ctx, p), "ampli_dev_free");
            }
        } dev{api, nullptr, {}};
        int device = 0;
        if (const char *e = getenv("AMPLISOLVE_DEVICE")) device = atoi(e);
        dev.check(api->ctx_create(device, nullptr, &dev.ctx), "ampli_ctx_create");

        // batches of whole BGZF blocks, ~batch_bytes of records each; two pinned buffers: batch k+1 is inflated while batch k
        // is uploaded and counted
        size_t batch_bytes = (size_t)256 << 20;
        if (const char *e = getenv("AMPLISOLVE_BAM_BATCH_BYTES")) batch_bytes = (size_t)std::max(70000ll, atoll(e)); // tests: many small batches
        PinnedBuf hbuf[2];
        std::vector<uint64_t> offs[2];
        void *d_bam[2] = {nullptr, nullptr}, *d_off[2] = {nullptr, nullptr};
        size_t d_bam_cap[2] = {0, 0}, d_off_cap[2] = {0, 0};
        void *ev[2] = {nullptr, nullptr};
        for (int i = 0; i < 2; ++i) dev.check(api->event_create(&ev[i]), "ampli_event_create");
        struct EvGuard { const HipApi *api; void **ev; ~EvGuard() { for (int i = 0; i < 2; ++i) if (ev[i]) api->event_destroy(ev[i]); } } evg{api, ev};
        bool busy[2] = {false, false};

        Header hdr;
        bool have_header = false;
        uint64_t *d_keys = nullptr;
        int32_t *d_counts = nullptr;
        uint64_t *d_stats = nullptr;
        std::vector<uint64_t> keys;
        int64_t P = 0, n_records = 0, malformed = 0;

        std::vector<unsigned char> carry; // the incomplete record (or header) at the end of the previous batch
        uint64_t stream_pos = 0;          // uncompressed bytes inflated so far (for messages)
        size_t b0 = 0;
        int slot = 0;
        while (b0 < blocks.size()) {
            size_t b1 = b0, bytes = 0;
            while (b1 < blocks.size() && (b1 == b0 || bytes + blocks[b1].isize <= batch_bytes)) bytes += blocks[b1++].isize;
            if (busy[slot]) { // the slot's previous upload and kernel are done
                const double w0 = now();
                dev.check(api->event_sync(ev[slot]), "ampli_event_sync");
                t_wait += now() - w0;
                busy[slot] = false;
            }
            PinnedBuf &hb = hbuf[slot];
            hb.ensure(api, carry.size() + bytes + 16);
            if (!carry.empty()) memcpy(hb.p, carry.data(), carry.size());
            const double i0 = now();
            inflate_range(f, blocks, b0, b1, hb.p + carry.size(), threads, a.bam);
            t_inflate += now() - i0;
            const size_t n = carry.size() + bytes;
            size_t from = 0;
            if (!have_header) {
                if (!parse_header(hb.p, n, hdr, a.bam)) { // header longer than this batch: keep everything and read on
                    carry.assign(hb.p, hb.p + n);
                    stream_pos += bytes;
                    b0 = b1;
                    if (b0 >= blocks.size()) throw Error{AMPLI_E_INVALID, a.bam + ": truncated BAM header"};
                    continue;
                }
                have_header = true;
                from = hdr.end;
                // the panel as sorted unique keys (BAM reference id << 32 | position)
                std::unordered_map<std::string, int> rid;
                for (size_t i = 0; i < hdr.ref_names.size(); ++i) rid.emplace(hdr.ref_names[i], (int)i);
                for (auto &v : lines) {
                    auto it = rid.find(v.chrom);
                    if (it != rid.end() && v.pos < (1ll << 31)) keys.push_back(((uint64_t)(uint32_t)it->second << 32) | (uint64_t)v.pos);
                }
                std::sort(keys.begin(), keys.end());
                keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
                P = (int64_t)keys.size();
                for (auto &v : lines) {
                    auto it = rid.find(v.chrom);
                    if (it == rid.end() || v.pos >= (1ll << 31)) continue;
                    const uint64_t k = ((uint64_t)(uint32_t)it->second << 32) | (uint64_t)v.pos;
                    v.key_index = (int64_t)(std::lower_bound(keys.begin(), keys.end(), k) - keys.begin());
                }
                if (P > 0) {
                    d_keys = (uint64_t *)dev.alloc((size_t)P * 8);
                    d_counts = (int32_t *)dev.alloc((size_t)P * 8 * sizeof(int32_t));
                    d_stats = (uint64_t *)dev.alloc(16);
                    dev.check(api->copy_h2d(dev.ctx, d_keys, keys.data(), (size_t)P * 8), "ampli_copy_h2d");
                    dev.check(api->memset_d(dev.ctx, d_counts, 0, (size_t)P * 8 * sizeof(int32_t)), "ampli_memset_d");
                    dev.check(api->memset_d(dev.ctx, d_stats, 0, 16), "ampli_memset_d");
                }
            }
            offs[slot].clear();
            const size_t done = scan_records(hb.p, from, n, offs[slot], malformed, a.bam, stream_pos - carry.size());
            stream_pos += bytes;
            carry.assign(hb.p + done, hb.p + n);
            n_records += (int64_t)offs[slot].size();
            if (P > 0 && !offs[slot].empty()) {
                // (the slot's previous kernel has finished: its event was waited for above)
                // (+64: the staged kernel copies whole 16-byte pieces, so up to 15 bytes behind the last record are read)
                if (d_bam_cap[slot] < done + 64) { dev.release(d_bam[slot]); d_bam[slot] = dev.alloc(done + done / 8 + 64); d_bam_cap[slot] = done + done / 8 + 64; }
                const size_t ob = offs[slot].size() * 8;
                if (d_off_cap[slot] < ob) { dev.release(d_off[slot]); d_off[slot] = dev.alloc(ob + ob / 8); d_off_cap[slot] = ob + ob / 8; }
                dev.check(api->copy_h2d(dev.ctx, d_bam[slot], hb.p, done), "ampli_copy_h2d");
                dev.check(api->copy_h2d(dev.ctx, d_off[slot], offs[slot].data(), ob), "ampli_copy_h2d");
                dev.check(api->pileup_count(dev.ctx, (const uint8_t *)d_bam[slot], (const uint64_t *)d_off[slot], (int64_t)offs[slot].size(), d_keys, P,
                                            a.mbq, a.mrq, d_counts, d_stats), "ampli_pileup_count");
                dev.check(api->event_record(dev.ctx, ev[slot]), "ampli_event_record");
                busy[slot] = true;
            }
            b0 = b1;
            slot ^= 1;
        }
        if (!have_header) throw Error{AMPLI_E_INVALID, a.bam + ": truncated BAM header"};
        if (!carry.empty()) // a cut-off file or a desynchronised stream: no counts file is written from part of the reads
            throw Error{AMPLI_E_INVALID, a.bam + ": " + std::to_string(carry.size()) + " bytes behind the last complete alignment record (truncated or corrupt file)"};

        std::vector<int32_t> counts((size_t)P * 8, 0);
        uint64_t st[2] = {0, 0};
        if (P > 0) {
            dev.check(api->copy_d2h(dev.ctx, counts.data(), d_counts, counts.size() * sizeof(int32_t)), "ampli_copy_d2h");
            dev.check(api->copy_d2h(dev.ctx, st, d_stats, 16), "ampli_copy_d2h");
        }
        dev.check(api->sync(dev.ctx), "ampli_sync");

        // <out>/<bam name without .bam>.PILEUP.ASEQ, one line per listed position whose depth reaches mdc, in the list's order
        std::string base = a.bam;
        const size_t sl = base.rfind('/');
        if (sl != std::string::npos) base = base.substr(sl + 1);
        if (base.size() > 4 && base.compare(base.size() - 4, 4, ".bam") == 0) base.resize(base.size() - 4);
        const std::string out_path = a.out_dir + "/" + base + ".PILEUP.ASEQ";
        FILE *o = fopen(out_path.c_str(), "w");
        if (!o) throw Error{AMPLI_E_INVALID, "cannot write " + out_path};
        fputs("chr\tpos\tdbsnp\tMAF\tref\talt\tA\tC\tG\tT\tRD\tArs\tCrs\tGrs\tTrs\n", o);
        int64_t written = 0;
        for (const auto &v : lines) {
            static const int32_t zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            const int32_t *c = v.key_index >= 0 ? &counts[(size_t)v.key_index * 8] : zero;
            const long long rd = (long long)c[0] + c[1] + c[2] + c[3];
            if (rd < a.mdc) continue;
            fprintf(o, "%s\t%lld\t%s\t.\t%s\t%s\t%d\t%d\t%d\t%d\t%lld\t%d\t%d\t%d\t%d\n", v.chrom.c_str(), (long long)v.pos, v.id.c_str(), v.ref.c_str(),
                    v.alt.c_str(), c[0], c[1], c[2], c[3], rd, c[4], c[5], c[6], c[7]);
            ++written;
        }
        if (fclose(o) != 0) throw Error{AMPLI_E_INVALID, "cannot write " + out_path};
        std::cout << "computeCounts (MI355X-native build): " << a.bam << "\n\t" << n_records << " alignment records (" << malformed << " malformed), "
                  << st[0] << " kept (mrq " << a.mrq << "), " << st[1] << " bases counted (mbq " << a.mbq << ") over " << lines.size() << " listed positions\n\t"
                  << written << " lines with depth >= " << a.mdc << " written to " << out_path << std::endl;
        if (getenv("AMPLISOLVE_TIMING")) {
            size_t total = 0;
            for (auto &b : blocks) total += b.isize;
            std::cerr << "TIMING total " << now() - t_start << " inflate " << t_inflate << " (" << threads << " threads, " << total << " bytes) device_wait " << t_wait
                      << std::endl;
        }
        if (a.stats) { a.stats[0] = n_records; a.stats[1] = (int64_t)st[0]; a.stats[2] = (int64_t)st[1]; a.stats[3] = written; }
        return 0;
    } catch (const Error &e) {
        std::cout << "computeCounts: " << e.msg << std::endl;
        if (a.error) *a.error = e.msg;
        return e.code ? e.code : -1;
    }
}

} // namespace ampli
