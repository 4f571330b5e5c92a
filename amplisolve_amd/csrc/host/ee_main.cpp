// AmpliSolveErrorEstimation -- drop-in command line (EE:241-520).
//   AmpliSolveErrorEstimation panel_design=<bed> reference_genome=<fa> germline_dir=<dir|not_available>
//                             C_value=<f> coverage_cutoff=<i> default_error=<f> output_dir=<dir>
// Exactly 7 key=value tokens in this order (EE:266, EE:300-326).  Like the reference, the process exits 0 on
// every path (EE:272, EE:3057-3074); set AMPLISOLVE_STRICT_EXIT=1 to get a non-zero status on failure.
#include <clocale>
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "host.hpp"

static std::string token(const char *arg, const char *key)
{
    char buf[4096];
    buf[0] = 0;
    std::string fmt = std::string(key) + "=%4000s"; // sscanf(argv[i], "key=%s", ...) (EE:300-326)
    sscanf(arg, fmt.c_str(), buf);
    return buf;
}

int main(int argc, char **argv)
{
    const double t_main = ampli::PhaseClock::now();
    setlocale(LC_ALL, "");
    const bool strict = getenv("AMPLISOLVE_STRICT_EXIT") != nullptr;
    if (argc != 8) {
        std::cout << "************************************************************************************************************************************" << std::endl;
        std::cout << "                                        Your input arguments are not correct !" << std::endl;
        std::cout << "Usage:\n\tAmpliSolveErrorEstimation panel_design=<bed> reference_genome=<fasta> germline_dir=<dir|not_available> "
                     "C_value=<float> coverage_cutoff=<int> default_error=<float> output_dir=<dir>\n\tAll arguments are required, in this order." << std::endl;
        return strict ? 2 : 0;
    }
    ampli::EeArgs a;
    a.panel_design = token(argv[1], "panel_design");
    a.reference_genome = token(argv[2], "reference_genome");
    a.germline_dir = token(argv[3], "germline_dir");
    a.C_value = token(argv[4], "C_value");
    a.coverage_cutoff = token(argv[5], "coverage_cutoff");
    a.default_error = token(argv[6], "default_error");
    a.output_dir = token(argv[7], "output_dir");
    if (const char *e = getenv("AMPLISOLVE_REFBASES_FILE")) a.refbases_file = e; // pre-computed chrom/pos/base table instead of the FASTA
    a.native = ampli::native_dist_from_env(a.output_dir); // AMPLISOLVE_WORLD_SIZE > 1: one shard of a one-process-per-GPU run (RCCL)
    a.process_ends = !(getenv("AMPLISOLVE_EXIT") && std::string(getenv("AMPLISOLVE_EXIT")) == "orderly"); // finish_process() leaves with _exit
    const int rc = ampli::run_error_estimation(a);
    if (getenv("AMPLISOLVE_TIMING")) ampli::PhaseClock::report(std::cerr, ampli::PhaseClock::now() - t_main);
    const int status = (strict || a.native.world > 1) ? (rc ? 1 : 0) : 0; // a failed shard must be visible to whatever launched the shards
    ampli::finish_process(status); // every file is written and closed: leave without the runtime's orderly teardown (AMPLISOLVE_EXIT=orderly keeps it)
    return status;
}
