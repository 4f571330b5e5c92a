// oracle/ref_vc_driver.cpp -- TEST INFRASTRUCTURE (reference build driver), not product code.
//
// Links against oracle/_ref/vc_host_ref.o, which oracle/Makefile compiles from the Boost-free lines of
// /root/reference/source_codes/AmpliSolveVariantCalling.cpp WHERE THEY LIE (sed line ranges piped into g++; the
// source is never copied and nothing stands in for Boost: callVariants / fisherTest stay undefined and uncalled).
// Every value this driver prints is produced by the reference's own functions on the reference's own global maps:
//   storeInputFile   VC:430-576   storeCountList   VC:580-627   generateCountList VC:387-394
//   find_kmer_down   VC:3307-3458 find_kmer_up     VC:3461-3613 homopolymerTest   VC:3615-3718
//
// usage: vc_ref_driver maps <error_table> <dummy_vcf_out>
//   storeInputFile as main() calls it (VC:320), then the four maps it fills, one line per entry, sorted by key
//   (the maps are unordered and nothing downstream iterates them):
//     R <chr_pos> <reference cell>         ReferenceBase_Hash        (VC:505)
//     D <chr_pos> <position>               DuplicatePosition_Hash    (VC:508-512)
//     T <chr_pos_X> <threshold cell>       Thresholds_Hash_Analytic  (VC:519-538)
//     G <chr_pos_X> <germ-max cell>        Germline_Max_Hash         (VC:541-560)
//   <dummy_vcf_out> is the by-product file storeInputFile itself writes (VC:573).
// usage: vc_ref_driver context <error_table> <dummy_vcf_out>
//   storeInputFile, then for every data row of the table in file order (first row of a position only; key as VC:828):
//     C <chrom> <position> <find_kmer_down> <find_kmer_up> <homopolymerTest for sub = A> <C> <G> <T>
//   with the arguments callVariants passes (VC:964-965, 1017: ReferenceBase_Hash, chrom, the integer position).
// usage: vc_ref_driver time <error_table> <dummy_vcf_out>
//   storeInputFile alone, timed; on stderr: "TIMING storeInputFile <seconds> positions <n> thresholds <n>"
// usage: vc_ref_driver order <tumour_dir> <list_file_out>
//   generateCountList + storeCountList as main() calls them (VC:328-333), then the iteration order of
//   TumourFileList_Hash -- the order callVariants visits the files in (VC:672) -- as "<sample name>\t<listed path>".
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

// declarations mirror VC:156-158, 171-173 and the globals of VC:178-197
void storeInputFile(char *file_name, char *dummyVCF);
void storeCountList(char *list_name, char *COUNT_DIR, std::unordered_map<std::string, std::string> &Hash);
void generateCountList(char *dir_path, char *list_name);
void find_kmer_down(std::unordered_map<std::string, std::string> &Hash, char *chromosome, int position, char *output);
void find_kmer_up(std::unordered_map<std::string, std::string> &Hash, char *chromosome, int position, char *output);
int homopolymerTest(char *down, char *up, char sub);
extern std::unordered_map<std::string, std::string> ReferenceBase_Hash;
extern std::unordered_map<std::string, std::string> DuplicatePosition_Hash;
extern std::unordered_map<std::string, std::string> Thresholds_Hash_Analytic;
extern std::unordered_map<std::string, std::string> Germline_Max_Hash;
extern std::unordered_map<std::string, std::string> TumourFileList_Hash;

static void dump_sorted(const char *tag, const std::unordered_map<std::string, std::string> &m)
{
    std::vector<std::pair<std::string, std::string>> v(m.begin(), m.end());
    std::sort(v.begin(), v.end());
    for (const auto &kv : v) std::cout << tag << " " << kv.first << " " << kv.second << "\n";
}

int main(int argc, char **argv)
{
    if (argc != 4) {
        fprintf(stderr, "usage: vc_ref_driver maps|context|time <error_table> <dummy_vcf_out> | order <tumour_dir> <list_file_out>\n");
        return 2;
    }
    const std::string mode = argv[1];
    // the reference prints its progress on stdout; keep ours apart by sending the reference's chatter to stderr
    std::streambuf *ours = std::cout.rdbuf();
    std::cout.rdbuf(std::cerr.rdbuf());
    if (mode == "maps" || mode == "context") {
        storeInputFile(argv[2], argv[3]);
        std::cout.rdbuf(ours);
        if (mode == "maps") {
            dump_sorted("R", ReferenceBase_Hash);
            dump_sorted("D", DuplicatePosition_Hash);
            dump_sorted("T", Thresholds_Hash_Analytic);
            dump_sorted("G", Germline_Max_Hash);
            return 0;
        }
        // the table's positions in file order, read from the by-product file the reference just wrote (chrom \t position ...)
        std::ifstream vcf(argv[3]);
        std::string line;
        std::set<std::string> seen;
        while (std::getline(vcf, line)) {
            char chrom[1000], position[1000];
            if (sscanf(line.c_str(), "%999s %999s", chrom, position) != 2) continue;
            if (!seen.insert(std::string(chrom) + "_" + position).second) continue;
            char down[1000], up[1000];
            memset(down, 0, sizeof down);
            memset(up, 0, sizeof up);
            const int position_prompt_key = atoi(position); // VC:828 forms the key from the text; the k-mer calls take the integer
            find_kmer_down(ReferenceBase_Hash, chrom, position_prompt_key, down);
            find_kmer_up(ReferenceBase_Hash, chrom, position_prompt_key, up);
            std::cout << "C " << chrom << " " << position << " " << down << " " << up;
            for (const char sub : {'A', 'C', 'G', 'T'}) std::cout << " " << homopolymerTest(down, up, sub);
            std::cout << "\n";
        }
        return 0;
    }
    if (mode == "time") { // wall time of the reference's own table load (VC:430-576), for bench.py's cpu_baseline
        const auto t0 = std::chrono::steady_clock::now();
        storeInputFile(argv[2], argv[3]);
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::cout.rdbuf(ours);
        // on stderr, where bench.py reads every TIMING line of a child from
        fprintf(stderr, "TIMING storeInputFile %.6f positions %zu thresholds %zu\n", s, ReferenceBase_Hash.size(), Thresholds_Hash_Analytic.size());
        return 0;
    }
    if (mode == "order") {
        generateCountList(argv[2], argv[3]);
        storeCountList(argv[3], argv[2], TumourFileList_Hash);
        std::cout.rdbuf(ours);
        for (auto it = TumourFileList_Hash.begin(); it != TumourFileList_Hash.end(); ++it) std::cout << it->second << "\t" << it->first << "\n";
        return 0;
    }
    fprintf(stderr, "unknown mode %s\n", argv[1]);
    return 2;
}
