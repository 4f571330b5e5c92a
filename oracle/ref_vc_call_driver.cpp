// oracle/ref_vc_call_driver.cpp -- TEST INFRASTRUCTURE (reference build driver), not product code.
//
// "callVariants without its Fisher statements".  Links against oracle/_ref/vc_call_ref.o, which oracle/Makefile compiles
// from /root/reference/source_codes/AmpliSolveVariantCalling.cpp WHERE IT LIES: the whole translation unit piped through
// `sed` into g++ with -Dmain=vc_ref_main, minus exactly
//   VC:135                    #include <boost/math/distributions/hypergeometric.hpp>   (Boost.Math is absent here)
//   VC:3797-3814              fisherTest's body (the only user of Boost)
//   VC:902, 1100, 1296, 1499, 1697, 1896, 2102, 2303, 2504, 2712, 2916, 3116
//                             the 12 statements `p=fisherTest(...);`, each preceded by `p=-1;` (VC:901 ...)
// Nothing stands in for Boost and no line is rewritten: with those statements gone `p` keeps the -1 of the line before, so
// in this build every emitted call prints FisherPvalue = -1 and carries the Fisher flag YES (-1 <= p_value, VC:903-906).
// Every other byte of Summary_Variant_Info.txt and of the <sample>.vcf files -- the per-line gate (VC:752-898 and its 11
// clones), the VAF columns, the sticky stream precision, the flag set coming out of an unordered_map in its own order,
// the C->G "-" ID, the file order -- is produced by the reference's own callVariants (VC:633-3304), reached through the
// reference's own main() (VC:199-360) in oracle/_ref/AmpliSolveVariantCalling_noFisher (the same lines built as a program:
//   AmpliSolveVariantCalling_noFisher errorFile=<t> tumour_dir=<d> output_dir=<o> coverage_cutoff=<n> p_value=<p>
//   -- main() copies errorFile into a 50-char buffer, VC:316: run it in the working directory with short names)
// or, here, in main()'s order with a clock around each step (main() itself is renamed away and never called).
//
// usage: vc_call_ref_driver time <error_table> <tumour_dir> <output_dir> <coverage_cutoff> <p_value>
//   storeInputFile -> generateCountList -> storeCountList -> callVariants as main() calls them (VC:320-344), wall seconds of each
//   on stderr as "TIMING <phase> <seconds>"; output_dir and output_dir/AmpliSolveVariantCalling_interm_files must exist.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <unordered_map>

// declarations mirror VC:156-173 and the globals of VC:178-197
void storeInputFile(char *file_name, char *dummyVCF);
void storeCountList(char *list_name, char *COUNT_DIR, std::unordered_map<std::string, std::string> &Hash);
void generateCountList(char *dir_path, char *list_name);
void callVariants(std::unordered_map<std::string, std::string> &ReferenceBase, std::unordered_map<std::string, std::string> &DuplicatePosition,
                  std::unordered_map<std::string, std::string> &TumourFileList, std::unordered_map<std::string, std::string> &Thresholds,
                  char *output_dir, char *tumourFile, int CovCut, float myPvalue);
extern std::unordered_map<std::string, std::string> ReferenceBase_Hash;
extern std::unordered_map<std::string, std::string> DuplicatePosition_Hash;
extern std::unordered_map<std::string, std::string> Thresholds_Hash_Analytic;
extern std::unordered_map<std::string, std::string> TumourFileList_Hash;
extern std::unordered_map<std::string, std::string> Germline_Hash_forPatients;

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    if (argc == 7 && std::string(argv[1]) == "time") {
        std::streambuf *ours = std::cout.rdbuf();
        std::cout.rdbuf(std::cerr.rdbuf()); // the reference's progress lines
        char dummy[1000], list[1000], summary[1000];
        snprintf(dummy, sizeof dummy, "%s/AmpliSolveVariantCalling_interm_files/dummyVCF_1.vcf", argv[4]);           // VC:313
        snprintf(list, sizeof list, "%s/AmpliSolveVariantCalling_interm_files/1_tumour_count_list_original.txt", argv[4]); // VC:328
        snprintf(summary, sizeof summary, "%s/Summary_Variant_Info.txt", argv[4]);                                    // VC:340
        const double t0 = now_s();
        storeInputFile(argv[2], dummy);
        const double t1 = now_s();
        generateCountList(argv[3], list);
        Germline_Hash_forPatients.insert(std::make_pair(std::string("test123456"), std::string("test123456"))); // VC:334-335
        storeCountList(list, argv[3], TumourFileList_Hash);
        const double t2 = now_s();
        int cov = atoi(argv[5]);
        if (cov <= 0) cov = 100; // VC:273-277
        float pv = (float)atof(argv[6]);
        if (pv <= 0 || pv > 1) pv = 0.05; // VC:286-290
        callVariants(ReferenceBase_Hash, DuplicatePosition_Hash, TumourFileList_Hash, Thresholds_Hash_Analytic, argv[4], summary, cov, pv);
        const double t3 = now_s();
        std::cout.rdbuf(ours);
        fprintf(stderr, "TIMING storeInputFile %.6f\nTIMING storeCountList %.6f\nTIMING callVariants %.6f files %zu\n", t1 - t0, t2 - t1, t3 - t2, TumourFileList_Hash.size());
        return 0;
    }
    fprintf(stderr, "usage: vc_call_ref_driver time <error_table> <tumour_dir> <output_dir> <coverage_cutoff> <p_value>\n");
    return 2;
}
