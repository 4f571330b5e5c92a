// AmpliSolveVariantCalling -- drop-in command line (VC:199-360).
//   AmpliSolveVariantCalling errorFile=<table> tumour_dir=<dir> output_dir=<dir> coverage_cutoff=<i> p_value=<f>
// Exactly 5 key=value tokens in this order (VC:216, VC:242-260).  Exit status as for AmpliSolveErrorEstimation.
#include <clocale>
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "host.hpp"

static std::string token(const char *arg, const char *key)
{
    char buf[4096];
    buf[0] = 0;
    std::string fmt = std::string(key) + "=%4000s";
    sscanf(arg, fmt.c_str(), buf);
    return buf;
}

int main(int argc, char **argv)
{
    const double t_main = ampli::PhaseClock::now();
    setlocale(LC_ALL, "");
    const bool strict = getenv("AMPLISOLVE_STRICT_EXIT") != nullptr;
    if (argc != 6) {
        std::cout << "************************************************************************************************************************************" << std::endl;
        std::cout << "                                        Your input arguments are not correct !" << std::endl;
        std::cout << "Usage:\n\tAmpliSolveVariantCalling errorFile=<positionSpecificNoise table> tumour_dir=<dir> output_dir=<dir> "
                     "coverage_cutoff=<int> p_value=<float>\n\tAll arguments are required, in this order." << std::endl;
        return strict ? 2 : 0;
    }
    ampli::VcArgs a;
    a.error_file = token(argv[1], "errorFile");
    a.tumour_dir = token(argv[2], "tumour_dir");
    a.output_dir = token(argv[3], "output_dir");
    a.coverage_cutoff = token(argv[4], "coverage_cutoff");
    a.p_value = token(argv[5], "p_value");
    a.native = ampli::native_dist_from_env(a.output_dir); // AMPLISOLVE_WORLD_SIZE > 1: one shard of a one-process-per-GPU run (RCCL)
    a.process_ends = !(getenv("AMPLISOLVE_EXIT") && std::string(getenv("AMPLISOLVE_EXIT")) == "orderly"); // finish_process() leaves with _exit
    const int rc = ampli::run_variant_calling(a);
    if (getenv("AMPLISOLVE_TIMING")) ampli::PhaseClock::report(std::cerr, ampli::PhaseClock::now() - t_main);
    const int status = (strict || a.native.world > 1) ? (rc ? 1 : 0) : 0; // a failed shard must be visible to whatever launched the shards
    ampli::finish_process(status); // every file is written and closed: leave without the runtime's orderly teardown (AMPLISOLVE_EXIT=orderly keeps it)
    return status;
}
