// computeCounts -- BAM -> <out>/<sample>.PILEUP.ASEQ, the pre-processing step of AmpliSolve
// (/root/reference/Execution_examples.md:16-46; upstream: a source-less binary, "a simplified version of ASEQ" in PILEUP mode).
//   computeCounts vcf=<positions> bam=<file.bam> threads=<int> mbq=<int> mrq=<int> mdc=<int> out=<dir>
// key=value tokens as in the reference's usage line; vcf, bam and out are required, the rest default to 4 / 20 / 20 / 20
// (the triplet the reference recommends for Ion AmpliSeq data).  Exit status: 0 on success, 1 on failure.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>

#include "host.hpp"

int main(int argc, char **argv)
{
    ampli::CcArgs a;
    bool bad = argc < 2;
    for (int i = 1; i < argc; ++i) {
        const char *eq = strchr(argv[i], '=');
        if (!eq) { bad = true; continue; }
        const std::string key(argv[i], eq - argv[i]), val(eq + 1);
        if (key == "vcf") a.vcf = val;
        else if (key == "bam") a.bam = val;
        else if (key == "out") a.out_dir = val;
        else if (key == "threads") a.threads = atoi(val.c_str());
        else if (key == "mbq") a.mbq = atoi(val.c_str());
        else if (key == "mrq") a.mrq = atoi(val.c_str());
        else if (key == "mdc") a.mdc = atoi(val.c_str());
        else bad = true;
    }
    if (bad || a.vcf.empty() || a.bam.empty() || a.out_dir.empty()) {
        std::cout << "Usage:\n\tcomputeCounts vcf=<dummyVCF.txt> bam=<file.bam> [threads=<int>] [mbq=<int>] [mrq=<int>] [mdc=<int>] out=<dir>\n"
                     "\tvcf: every position of the panel, one per line (chr pos ...); mbq / mrq / mdc: minimum base quality, read (mapping)\n"
                     "\tquality and depth of coverage (20 20 20 recommended); writes <out>/<bam name>.PILEUP.ASEQ" << std::endl;
        return 1;
    }
    return ampli::run_compute_counts(a) ? 1 : 0;
}
