// oracle/ref_ee_driver.cpp -- TEST INFRASTRUCTURE (reference build driver), not product code.
//
// Links against oracle/_ref/ee_ref.o, which oracle/Makefile compiles from
// /root/reference/source_codes/AmpliSolveErrorEstimation.cpp WHERE IT LIES
// (-Dmain=ee_ref_main; the source is never copied).  This driver calls the
// reference's own external-linkage functions in the order its main() does
// (EE:426-454), skipping only generateReferenceBases (EE:578-670: one
// `samtools faidx` popen per panel position; samtools is absent here and that
// step is not on the hot path).  The reference-base and duplicate tables it
// would have produced are passed in as files in the formats storeReference
// (EE:963) and storeDuplicates (EE:1012) read.
//
// usage: ee_ref_driver --default <bed> <refbases.txt> <dups.txt> <default_error> <out_dir>
//   the germline_dir=not_available branch of main() (EE:472-506): storeReference, storeDuplicates, then the reference's
//   generateFinalOutput_default (EE:2948-3043) writes <out_dir>/positionSpecificNoise_default.txt.  default_error is
//   passed as main() would after its own conversion (EE:353-363: atof, and 0.01 for a value <= 0).
// usage: ee_ref_driver <bed> <refbases.txt> <dups.txt> <germline_dir> <C> <cov> <out_dir> [dump_prefix]
//   writes <out_dir>/positionSpecificNoise_<C>.txt exactly as the reference does; with dump_prefix also
//   <dump_prefix>.order (sample visit order), <dump_prefix>.counts (Count_Hash: the integer quorum
//   counts, EE:1665), <dump_prefix>.walk (per key the records in the order estimateThresholds adds them: its
//   equal_range walk) and prints per-phase wall seconds on stderr as "TIMING <phase> <seconds>".
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <unordered_map>

// declarations mirror EE:163-239
void storeReference(char *reference_bases_name, std::unordered_map<std::string, std::string> &Hash);
void storeDuplicates(char *amplicon_positions, std::unordered_map<std::string, std::string> &Hash);
void storeGermlineStatistics(std::unordered_map<std::string, std::string> &FILE_Hash,
                             std::unordered_multimap<std::string, std::string> &Value_Hash_Thres,
                             std::unordered_map<std::string, double> &Germline_M_Hash, int coverageCutoff);
void generateCountList(char *dir_path, char *list_name);
void storeCountList(char *list_name, char *COUNT_DIR, std::unordered_map<std::string, std::string> &Hash);
void estimateThresholds(float norm_factor, std::unordered_map<std::string, std::string> &Position_Hash,
                        std::unordered_multimap<std::string, std::string> &Value_Hash,
                        std::unordered_map<std::string, std::string> &Thresholds,
                        std::unordered_map<std::string, std::string> &CountHash,
                        std::unordered_map<std::string, std::string> &RatioHash, int coverageCutoff);
void generateFinalOutput(float C_value_float, char *panelDesign,
                         std::unordered_map<std::string, std::string> &Reference_Hash,
                         std::unordered_map<std::string, std::string> &Duplicate_Hash,
                         std::unordered_map<std::string, std::string> &Thresholds,
                         std::unordered_map<std::string, double> &Germline_M_Hash, char *output_dir,
                         std::unordered_map<std::string, std::string> &CountHash,
                         std::unordered_map<std::string, std::string> &RatioHash);

void generateFinalOutput_default(float C_value_float, char *panelDesign, std::unordered_map<std::string, std::string> &Reference_Hash,
                                 std::unordered_map<std::string, std::string> &Duplicate_Hash, char *output_dir, float defaultError);

extern std::unordered_map<std::string, std::string> GermlineCountFileList_Hash;
extern std::unordered_map<std::string, std::string> ReferenceBase_Hash;
extern std::unordered_map<std::string, std::string> DuplicatePosition_Hash;
extern std::unordered_multimap<std::string, std::string> GermlineValues_Hash_forThresholds;
extern std::unordered_map<std::string, double> Germline_Max_Hash;
extern std::unordered_map<std::string, std::string> Count_Hash;
extern std::unordered_map<std::string, std::string> Ratio_Hash;
extern std::unordered_map<std::string, std::string> Thresholds_Hash_Analytic;

static double now_s()
{
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char **argv)
{
    if (argc == 7 && strcmp(argv[1], "--default") == 0) {
        storeReference(argv[3], ReferenceBase_Hash);      // EE:493
        storeDuplicates(argv[4], DuplicatePosition_Hash); // EE:500
        generateFinalOutput_default(0.0f, argv[2], ReferenceBase_Hash, DuplicatePosition_Hash, argv[6], (float)std::atof(argv[5])); // EE:501 (C is unused there)
        std::cout.flush();
        return 0;
    }
    if (argc < 8) {
        fprintf(stderr, "usage: %s <bed> <refbases> <dups> <germline_dir> <C> <cov> <out_dir> [dump_prefix]\n", argv[0]);
        return 2;
    }
    char *bed = argv[1], *refbases = argv[2], *dups = argv[3], *gdir = argv[4], *outdir = argv[7];
    float C = (float)std::atof(argv[5]);
    int cov = std::atoi(argv[6]);
    const char *dump = argc > 8 ? argv[8] : nullptr;

    storeReference(refbases, ReferenceBase_Hash);     // EE:426
    storeDuplicates(dups, DuplicatePosition_Hash);    // EE:433
    char list_name[2000];
    snprintf(list_name, sizeof list_name, "%s/ref_germline_count_list.txt", outdir);
    generateCountList(gdir, list_name);               // EE:443
    storeCountList(list_name, gdir, GermlineCountFileList_Hash); // EE:445

    double t0 = now_s();
    storeGermlineStatistics(GermlineCountFileList_Hash, GermlineValues_Hash_forThresholds, Germline_Max_Hash, cov); // EE:449
    double t1 = now_s();
    estimateThresholds(C, ReferenceBase_Hash, GermlineValues_Hash_forThresholds, Thresholds_Hash_Analytic,
                       Count_Hash, Ratio_Hash, cov); // EE:452
    double t2 = now_s();
    generateFinalOutput(C, bed, ReferenceBase_Hash, DuplicatePosition_Hash, Thresholds_Hash_Analytic,
                        Germline_Max_Hash, outdir, Count_Hash, Ratio_Hash); // EE:454
    double t3 = now_s();
    std::cout.flush();
    fprintf(stderr, "TIMING storeGermlineStatistics %.6f\n", t1 - t0);
    fprintf(stderr, "TIMING estimateThresholds %.6f\n", t2 - t1);
    fprintf(stderr, "TIMING generateFinalOutput %.6f\n", t3 - t2);
    fprintf(stderr, "TIMING records %zu\n", GermlineValues_Hash_forThresholds.size() / 4);

    if (dump) {
        std::string p(dump);
        std::ofstream o1(p + ".order");
        for (auto it = GermlineCountFileList_Hash.begin(); it != GermlineCountFileList_Hash.end(); ++it)
            o1 << it->second << "\n"; // visit order of EE:1081
        std::ofstream o2(p + ".counts");
        for (auto it = Count_Hash.begin(); it != Count_Hash.end(); ++it) o2 << it->first << "\t" << it->second << "\n";
        // the order estimateThresholds adds a key's records in (EE:1555, 1565: `equal_range` of the reference's own multimap, walked with
        // the reference's own container): one line per key, the values ("Xfw_FW_Xbw_BW", EE:1236-1245) in iteration order
        std::ofstream o3(p + ".walk");
        for (auto a = ReferenceBase_Hash.begin(); a != ReferenceBase_Hash.end(); ++a)
            for (const char *nt : {"A", "C", "G", "T"}) {
                const std::string key = a->first + "_" + nt;
                auto r = GermlineValues_Hash_forThresholds.equal_range(key);
                o3 << key;
                for (auto b = r.first; b != r.second; ++b) o3 << "\t" << b->second;
                o3 << "\n";
            }
    }
    return 0;
}
