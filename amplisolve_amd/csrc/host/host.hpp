// amplisolve_amd/csrc/host/host.hpp -- C++ host side of the two AmpliSolve command lines.
//
// Parsers, panel index, SoA packer, table reader/writer, annotation and the
// pipelines that call libamplisolve_hip.so.  No arithmetic of the hot path
// lives here.  EE:n / VC:n cite /root/reference/source_codes/AmpliSolve{ErrorEstimation,VariantCalling}.cpp.
#pragma once
#include <cstdint>
#include <iosfwd>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/amplisolve_hip.h"
#include "../../../include/amplisolve_host.h"

namespace ampli {

struct BedRow {
    std::string chrom;
    int start = -1, end = -1;
};

// The panel: unique positions in order of first appearance in the BED walk
// (rows in file order, start..end inclusive, 1-based: EE:633-649, EE:2595-2606).
struct Panel {
    std::vector<BedRow> rows;
    std::vector<std::string> chroms;                  // chrom id -> name
    std::unordered_map<std::string, int> chrom_id;    // name -> id
    std::vector<int32_t> pos_chrom, pos_coord;        // per unique position p
    std::vector<uint32_t> walk;                       // BED walk (duplicates repeated) -> p
    std::unordered_map<uint64_t, uint32_t> index;     // (chrom id << 32 | coord) -> p
    std::vector<std::string> ref_base;                // per p, as read (case preserved)
    std::vector<uint8_t> ref_code;                    // 0..3 = A,C,G,T (exact, upper case), 255 otherwise
    std::vector<uint8_t> dup;                         // per p: listed more than once (EE:657-664)
    // error-table columns, when the panel was loaded from positionSpecificNoise_*.txt (VC:430-576): the file's text with
    // every cell NUL-terminated in place, and per unique position the offsets of its 4 threshold + 4 germ-max cells
    std::string table_text;
    std::vector<uint32_t> cell_off[8];
    bool from_table() const { return !cell_off[0].empty() || (!table_text.empty() && pos_coord.empty()); }
    const char *thr_cell(int nt, int64_t p) const { return table_text.data() + cell_off[nt][(size_t)p]; }      // Thresholds_Hash_Analytic (VC:519-538)
    const char *germ_cell(int nt, int64_t p) const { return table_text.data() + cell_off[4 + nt][(size_t)p]; } // Germline_Max_Hash (VC:541-560)

    int64_t P() const { return (int64_t)pos_coord.size(); }
    int find(const std::string &chrom, int coord) const;
    int add_position(const std::string &chrom, int coord);
    int add_position_id(int chrom_id_, int coord); // the chromosome is already in `chroms`
    void set_ref(uint32_t p, const std::string &base);
};

// A line whose RD column differs from A+C+G+T (EE:1178-1181, VC:762-765).  The reference keeps using the column:
// as the denominator of the Germ_Max AF (EE:1229-1232) and, in the caller, as RD in AF = X/RD and in the forward depth
// RD - RD_reverse of the Poisson test (VC:814-817, VC:895).  Sparse side list next to the dense records.
struct Irregular {
    uint32_t sample;     // sample index inside the cohort / chunk
    uint32_t record;     // record slot in [0, P+E)
    uint32_t occurrence; // which line of that position in the file (0 = the primary record)
    int32_t rd;          // the RD column
};

// A directory of .PILEUP.ASEQ files packed into the record SoA of include/amplisolve_hip.h
struct Cohort {
    std::vector<std::string> paths;   // in visit order (EE:1081 / VC:672)
    std::vector<std::string> names;   // sample names (EE:831)
    int64_t P = 0, E = 0;
    int32_t *recs = nullptr;          // [S][P+E][8], pinned when the HIP library is loaded
    bool pinned = false;
    std::vector<uint32_t> dup_off;    // [P+1]
    std::vector<uint32_t> ext_pos;    // [E]
    std::vector<int32_t> line_no;     // [S][P+E] data-line index inside the sample's file, -1 absent
    std::vector<Irregular> irregular; // lines with RD != A+C+G+T
    int64_t n_lines = 0, n_offpanel = 0, n_irregular = 0, n_malformed = 0;
    int first_sample = 0, total_samples = 0; // this cohort = samples [first_sample, first_sample + S()) of the directory's visit order
    ~Cohort();
    int S() const { return (int)paths.size(); }
    int64_t R() const { return P + E; }
};

struct Error {
    int code;
    std::string msg;
};

// One upload unit of a streamed cohort: consecutive samples of the visit order, records already in the layout the
// kernels read (24-byte records; int32 when a count of the chunk needs it), extras in their own array with the
// chunk's own slot layout (ampli_records of include/amplisolve_hip.h describes exactly this).
struct Chunk {
    int slot = 0, index = 0;       // ring slot; chunk number
    int first = 0, n = 0;          // samples [first, first + n) of the stream's file list
    bool last = false;
    int layout = AMPLI_RECORDS_U24;
    int64_t P = 0, E = 0;
    void *prim = nullptr;          // [n][P] records; page-aligned host memory the parsers fill BEFORE the HIP runtime is up,
    size_t prim_cap = 0;           // pinned late (pin(): hipHostRegister) by the consumer once there is a context
    bool prim_pinned = false;
    void *ext = nullptr;           // [n][E] records
    size_t ext_cap = 0;
    bool ext_pinned = false;
    void pin(ampli_ctx *ctx);      // register prim / ext with the runtime if they are not yet (no-op without the HIP library)
    std::vector<uint32_t> dup_off; // [P+1]
    std::vector<uint32_t> ext_pos; // [E]
    std::vector<int32_t> line_prim, line_ext; // [n][P], [n][E] data-line index inside the sample's file (-1 absent); variant calling only
    std::vector<Irregular> irregular;
    int64_t n_lines = 0, n_offpanel = 0, n_irregular = 0, n_malformed = 0;
    Chunk() = default;
    Chunk(const Chunk &) = delete;
    Chunk &operator=(const Chunk &) = delete;
    ~Chunk();
};

// The cohort as a stream of chunks: a producer thread (with n_threads parser workers) fills a small ring of page-aligned
// buffers ahead of the consumer (they need no HIP call, so parsing starts while the runtime is still coming up; the
// consumer pins a buffer the first time it uploads from it), so parsing of chunk k+1 overlaps the upload and the kernels of chunk k and the host
// never holds more than n_slots chunks (replaces the parse loops of EE:1100-1149 / VC:699-752).
class ChunkStream {
public:
    ChunkStream(const Panel &panel, std::vector<std::pair<std::string, std::string>> files, int n_threads, bool keep_line_no,
                size_t chunk_bytes, int n_slots);
    ~ChunkStream();
    ChunkStream(const ChunkStream &) = delete;
    ChunkStream &operator=(const ChunkStream &) = delete;
    Chunk *next();            // next chunk in order, nullptr after the last; rethrows the producer's Error
    void release(Chunk *c);   // the chunk's buffers may be refilled
    void shutdown();          // stop the producer and free the ring now (the destructor does the same)
    void abandon();           // stop the producer and LEAVE the ring's buffers to the end of the process (a command line about to exit:
                              //  unpinning and unmapping half a gigabyte beside the writers costs them more than the exit does)
    int samples_per_chunk() const;
    int chunks() const;
    double parse_seconds() const; // producer time spent parsing so far
private:
    struct Impl;
    Impl *im;
};

// ---- pipeline.cpp: where a command line's wall time goes (printed with AMPLISOLVE_TIMING) ----
// Named spans, summed per name, from any thread.  `critical` spans lie on the main thread's path, so they add up to the
// wall time; the others (side-thread context start-up, parser threads) are reported as overlapped work beside them.
struct PhaseClock {
    static double now();                                   // seconds, steady clock
    static void add(const char *name, double seconds, bool critical);
    struct Scope {
        const char *name;
        bool critical;
        double t0;
        Scope(const char *n, bool c = true) : name(n), critical(c), t0(now()) {}
        ~Scope() { add(name, now() - t0, critical); }
    };
    static void reset();
    static void report(std::ostream &os, double wall);     // "TIMING2 <name> <seconds> critical|overlapped" lines + the unattributed rest
};
// ---- panel.cpp ----
void panel_from_bed(const std::string &bed_path, Panel &out);                   // throws Error
void panel_load_refbases_file(Panel &p, const std::string &path);               // chrom pos base per line (EE:963)
void panel_load_fasta(Panel &p, const std::string &fasta_path);                 // replaces `samtools faidx` (EE:644)
void panel_write_interm_files(const Panel &p, const std::string &dir, int seed);// EE:601, 657-664
// ---- aseq.cpp ----
std::vector<std::pair<std::string, std::string>> list_count_files(const std::string &dir, const std::string &list_file); // EE:552-559, 794-841
std::vector<std::pair<std::string, std::string>> shard_of_files(const std::vector<std::pair<std::string, std::string>> &files, int shard_index,
                                                                int shard_count, int *first);
size_t record_bytes(int layout);
void fill_absent(int layout, char *dst, size_t n_records);
// shard_index / shard_count: keep only that contiguous range of the visit order (multi-process runs)
void cohort_load(const Panel &panel, const std::string &dir, const std::string &list_file, int n_threads, bool keep_line_no,
                 bool print_irregular, Cohort &out, int shard_index = 0, int shard_count = 1);
// ---- table.cpp ----
std::string format_rate_cell(uint8_t code, float r_fw, float r_bw, bool is_ref);  // EE:1704, 2670-2688
std::string format_germ_cell(uint8_t present, float v);                           // EE:2807-2849
void write_error_table(const Panel &panel, const float *rate, const uint8_t *code, const float *germ_val,
                       const uint8_t *germ_present, const std::string &path);
void write_error_table_default(const Panel &panel, float default_error, const std::string &path); // EE:2948-3043
void panel_from_error_table(const std::string &path, const std::string &dummy_vcf, Panel &out, std::vector<float> &thr); // VC:430-576
// ---- hip_loader.cpp ----
struct HipApi; // function pointers of libamplisolve_hip.so
const HipApi *hip_api(std::string *why = nullptr); // nullptr when the library cannot be loaded
// ---- pipeline.cpp ----
// one process per GPU without Python: this process is shard `rank` of `world`, the exchange steps go over RCCL
// (ampli_comm_* of include/amplisolve_hip.h); rank 0 publishes the communicator id in id_file
struct NativeDist {
    int rank = 0, world = 1;
    std::string id_file;
    int timeout_s = 120;
};
// AMPLISOLVE_WORLD_SIZE / AMPLISOLVE_RANK / AMPLISOLVE_ID_FILE / AMPLISOLVE_RCCL_TIMEOUT (the executables' multi-GPU mode)
NativeDist native_dist_from_env(const std::string &output_dir);
struct EeArgs {
    std::string panel_design, reference_genome, germline_dir, output_dir;
    std::string C_value = "0.002", coverage_cutoff = "100", default_error = "0.01";
    std::string refbases_file; // test hook: skip the FASTA, read chrom/pos/base lines
    const ampli_host_shard *shard = nullptr; // one shard of a multi-process run (include/amplisolve_host.h)
    NativeDist native;                       // or: one shard with the library's own RCCL transport (shard == nullptr)
    bool process_ends = false;               // the caller is an executable that exits right after: big buffers are left to the exit
};
struct VcArgs {
    std::string error_file, tumour_dir, output_dir, coverage_cutoff = "100", p_value = "0.05";
    const ampli_host_shard *shard = nullptr;
    NativeDist native;
    bool process_ends = false;
};
// computeCounts (bam.cpp): one BAM file -> <out_dir>/<name>.PILEUP.ASEQ
struct CcArgs {
    std::string vcf, bam, out_dir;
    int threads = 4, mbq = 20, mrq = 20, mdc = 20; // Execution_examples.md:46 recommends 20-20-20
    int64_t *stats = nullptr;                      // optional [4]: records, reads kept, bases counted, lines written
    std::string *error = nullptr;
};
int run_compute_counts(const CcArgs &a);
// BGZF + BAM structure only (no GPU): stats[4] = alignment records, uncompressed bytes, references, malformed records
void bam_scan(const std::string &bam, int n_threads, int64_t stats[4]);
// End of a command line: all outputs are written and closed.  Flushes stdio / iostreams and leaves with _exit(status), i.e.
// without the static destructors and atexit handlers of the HIP runtime (queue / signal / pool teardown that only serves a
// process that lives on; the driver reclaims everything at exit either way).  AMPLISOLVE_EXIT=orderly returns instead.
void finish_process(int status);
int run_error_estimation(const EeArgs &a);
int run_variant_calling(const VcArgs &a);
// ---- annotate.cpp ----
double fisher_two_sided(int a, int b, int c, int d);                            // VC:3797-3814 (own hypergeometric pmf)
double fisher_two_sided_direct(int a, int b, int c, int d);                     // the same, every term from log-gamma (check)
long double score_reference_sequence(int k, int rd, float err);                 // VC:3834-3884, for calls within rounding of a gate
std::string kmer_down(const Panel &p, const std::string &chrom, int pos);       // VC:3307-3458
std::string kmer_up(const Panel &p, const std::string &chrom, int pos);         // VC:3461-3613
int homopolymer_test(const std::string &down, const std::string &up, char sub); // VC:3615-3718

} // namespace ampli
