// amplisolve_amd/csrc/host/panel.cpp -- BED panel, reference bases, duplicate positions.
#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "host.hpp"

namespace ampli {

int Panel::find(const std::string &chrom, int coord) const
{
    auto c = chrom_id.find(chrom);
    if (c == chrom_id.end()) return -1;
    auto it = index.find(((uint64_t)(uint32_t)c->second << 32) | (uint32_t)coord);
    return it == index.end() ? -1 : (int)it->second;
}

int Panel::add_position(const std::string &chrom, int coord)
{
    int cid;
    auto c = chrom_id.find(chrom);
    if (c == chrom_id.end()) {
        cid = (int)chroms.size();
        chroms.push_back(chrom);
        chrom_id.emplace(chrom, cid);
    } else {
        cid = c->second;
    }
    return add_position_id(cid, coord);
}

int Panel::add_position_id(int cid, int coord)
{
    const uint64_t key = ((uint64_t)(uint32_t)cid << 32) | (uint32_t)coord;
    const uint32_t p = (uint32_t)pos_coord.size();
    auto ins = index.emplace(key, p);
    if (!ins.second) return (int)ins.first->second;
    pos_chrom.push_back(cid);
    pos_coord.push_back(coord);
    ref_base.emplace_back();
    ref_code.push_back(255);
    dup.push_back(0);
    return (int)p;
}

void Panel::set_ref(uint32_t p, const std::string &base)
{
    ref_base[p] = base;
    // the reference compares with "A","C","G","T" exactly (EE:2668-2670, VC:869): case sensitive, whole string
    uint8_t c = 255;
    if (base == "A") c = 0;
    else if (base == "C") c = 1;
    else if (base == "G") c = 2;
    else if (base == "T") c = 3;
    ref_code[p] = c;
}

// Lines as `fscanf("%1000[^\n]\n")` + `sscanf("%s\t%d\t%d...")` see them (EE:615-633): blank lines vanish,
// CR is whitespace, columns beyond the third are ignored here.
void panel_from_bed(const std::string &bed_path, Panel &out)
{
    std::ifstream in(bed_path);
    if (!in) throw Error{AMPLI_E_INVALID, "Cannot open file: " + bed_path};
    std::string line;
    while (std::getline(in, line)) {
        char chrom[1024];
        int a = -1, b = -1;
        if (line.size() > 1000) line.resize(1000);
        if (sscanf(line.c_str(), "%1000s %d %d", chrom, &a, &b) < 1) continue; // whitespace-only line
        BedRow r;
        r.chrom = chrom;
        r.start = a;
        r.end = b;
        out.rows.push_back(r);
        for (int idx = a; idx <= b; ++idx) {
            const size_t before = out.pos_coord.size();
            const int p = out.add_position(r.chrom, idx);
            if (out.pos_coord.size() == before) out.dup[p] = 1; // listed again: `sort | uniq -d` (EE:663)
            out.walk.push_back((uint32_t)p);
        }
    }
}

void panel_load_refbases_file(Panel &p, const std::string &path)
{
    std::ifstream in(path);
    if (!in) throw Error{AMPLI_E_INVALID, "Cannot open " + path};
    std::string line;
    while (std::getline(in, line)) {
        char chrom[1024], pos[1024], base[1024];
        base[0] = 0;
        if (sscanf(line.c_str(), "%1000s %1000s %1000s", chrom, pos, base) < 2) continue; // EE:963
        const int i = p.find(chrom, atoi(pos));
        if (i >= 0 && p.ref_base[i].empty()) p.set_ref((uint32_t)i, base); // first insert wins (EE:971)
    }
}

namespace {
struct FaiEntry {
    int64_t length, offset, linebases, linewidth;
};

std::unordered_map<std::string, FaiEntry> load_or_build_fai(const std::string &fasta)
{
    std::unordered_map<std::string, FaiEntry> idx;
    std::ifstream fai(fasta + ".fai");
    if (fai) {
        std::string line;
        while (std::getline(fai, line)) {
            std::istringstream ss(line);
            std::string name;
            FaiEntry e;
            if (ss >> name >> e.length >> e.offset >> e.linebases >> e.linewidth) idx.emplace(name, e);
        }
        if (!idx.empty()) return idx;
    }
    // no index: one sequential scan of the FASTA (what `samtools faidx` does on first use)
    FILE *f = fopen(fasta.c_str(), "rb");
    if (!f) throw Error{AMPLI_E_INVALID, "Cannot open reference genome: " + fasta};
    std::vector<char> buf(1 << 22);
    std::string name;
    FaiEntry cur{0, 0, 0, 0};
    bool have = false, in_header = false, first_line = true;
    int64_t off = 0, line_bases = 0, line_start = 0;
    std::string hdr;
    size_t n;
    auto close_line = [&](int64_t width) {
        if (!have) return;
        if (first_line && line_bases > 0) {
            cur.linebases = line_bases;
            cur.linewidth = width;
            first_line = false;
        }
        cur.length += line_bases;
    };
    while ((n = fread(buf.data(), 1, buf.size(), f)) > 0) {
        for (size_t i = 0; i < n; ++i, ++off) {
            const char c = buf[i];
            if (in_header) {
                if (c == '\n') {
                    in_header = false;
                    size_t e = hdr.find_first_of(" \t\r");
                    name = hdr.substr(0, e);
                    cur = FaiEntry{0, off + 1, 0, 0};
                    have = true;
                    first_line = true;
                    line_bases = 0;
                    line_start = off + 1;
                } else {
                    hdr.push_back(c);
                }
            } else if (c == '>') {
                if (have) idx.emplace(name, cur);
                have = false;
                in_header = true;
                hdr.clear();
            } else if (c == '\n') {
                close_line(off + 1 - line_start);
                line_bases = 0;
                line_start = off + 1;
            } else if (c != '\r') {
                ++line_bases;
            }
        }
    }
    if (have) {
        close_line(off - line_start);
        idx.emplace(name, cur);
    }
    fclose(f);
    return idx;
}
} // namespace

// One pread per panel position instead of one `samtools faidx` process per position (EE:637-649).
void panel_load_fasta(Panel &p, const std::string &fasta_path)
{
    auto idx = load_or_build_fai(fasta_path);
    FILE *f = fopen(fasta_path.c_str(), "rb");
    if (!f) throw Error{AMPLI_E_INVALID, "Cannot open reference genome: " + fasta_path};
    for (int64_t i = 0; i < p.P(); ++i) {
        const std::string &chrom = p.chroms[p.pos_chrom[i]];
        auto it = idx.find(chrom);
        std::string base;
        if (it != idx.end() && p.pos_coord[i] >= 1 && p.pos_coord[i] <= it->second.length && it->second.linebases > 0) {
            const int64_t z = p.pos_coord[i] - 1;
            const int64_t o = it->second.offset + z / it->second.linebases * it->second.linewidth + z % it->second.linebases;
            char c = 0;
            if (fseeko(f, o, SEEK_SET) == 0 && fread(&c, 1, 1, f) == 1) base.assign(1, c);
        }
        p.set_ref((uint32_t)i, base);
    }
    fclose(f);
}

// <seed>_tmp_info.txt, _tmp_ref.txt, _tmp_ref_filtered.txt, _panelReferenceBases.txt, _ampliconDuplicatedPositions.txt
// with the reference's names and contents (EE:601, 644, 657-664).
void panel_write_interm_files(const Panel &p, const std::string &dir, int seed)
{
    const std::string pre = dir + "/" + std::to_string(seed) + "_";
    std::ofstream info(pre + "tmp_info.txt"), ref(pre + "tmp_ref.txt"), filt(pre + "tmp_ref_filtered.txt"),
        bases(pre + "panelReferenceBases.txt"), dups(pre + "ampliconDuplicatedPositions.txt");
    for (uint32_t w : p.walk) {
        const std::string &c = p.chroms[p.pos_chrom[w]];
        const int x = p.pos_coord[w];
        info << c << "\t" << x << "\n";
        ref << ">" << c << ":" << x << "-" << x << "\n" << p.ref_base[w] << "\n";
        filt << p.ref_base[w] << "\n";
        bases << c << "\t" << x << "\t" << p.ref_base[w] << "\n";
    }
    // `sort | uniq -d | cut -f1,2`: one line per duplicated position, in sort order of the full line
    std::vector<std::string> d;
    for (int64_t i = 0; i < p.P(); ++i)
        if (p.dup[i]) d.push_back(p.chroms[p.pos_chrom[i]] + "\t" + std::to_string(p.pos_coord[i]) + "\t" + p.ref_base[i]);
    std::sort(d.begin(), d.end(), [](const std::string &a, const std::string &b) { return strcoll(a.c_str(), b.c_str()) < 0; });
    for (auto &s : d) dups << s.substr(0, s.rfind('\t')) << "\n";
}

} // namespace ampli
