// amplisolve_amd/csrc/ampli_pileup.hip -- pileup_count_kernel + ampli_pileup_count (include/amplisolve_hip.h): the counting
// half of computeCounts (BAM -> .PILEUP.ASEQ); the container half is amplisolve_amd/csrc/host/bam.cpp.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "ampli_internal.h"

// ---------------------------------------------------------------------------
// pileup_count: the upstream step of the path, BAM alignments -> per-position base x strand counts (what ASEQ's PILEUP mode
// / the reference's binary-only computeCounts write as .PILEUP.ASEQ; /root/reference/Execution_examples.md:16-46).
// The host inflates the BGZF blocks and lists the byte offsets of the alignment records; this kernel decodes the records
// themselves.  One wave per read at a time, lanes over the bases of a CIGAR match run: the reads of an amplicon start at the
// same place, so a lane-per-read mapping would have all 64 lanes of a wave add to the SAME counter at every step.
//   kept reads: mapped, not secondary / QC-fail / duplicate (the pileup engine's default mask), MAPQ >= mrq;
//   counted bases: inside M / = / X runs (deletions and reference skips contribute nothing), A/C/G/T only, quality >= mbq;
//   counts[p][0..3] = A,C,G,T over both strands, counts[p][4..7] = the same on the reverse strand (flag 0x10).
// keys: the panel's unique positions as (BAM reference id << 32 | 1-based position), sorted ascending.
// ---------------------------------------------------------------------------
__device__ __forceinline__ unsigned ld_u32_unaligned(const unsigned char *p)
{
    unsigned v;
    __builtin_memcpy(&v, p, 4);
    return v;
}

// A workgroup takes PILEUP_READS consecutive reads (one wave per read at a time, lanes over the bases of a match run) and counts
// into a private LDS window of PILEUP_WINDOW panel positions that starts where its first read starts: a coordinate-sorted BAM keeps
// the reads of an amplicon together, so nearly every update is an LDS atomic and the window is flushed with one global atomic per
// non-zero counter (global atomics straight from the lanes ran at 4 G updates/s -- every read of an amplicon hits the same
// counters).  Updates outside the window (unsorted files, very long reads) go to the global counters directly.
constexpr int PILEUP_READS = 256;
constexpr int PILEUP_WINDOW = 1024;

__global__ __launch_bounds__(256) void pileup_count_kernel(const unsigned char *__restrict__ bam, const unsigned long long *__restrict__ rec_off,
                                                           const long long n_reads, const unsigned long long *__restrict__ keys, const long long P,
                                                           const int mbq, const int mrq, int *__restrict__ counts, unsigned long long *__restrict__ stats)
{
    __shared__ int win[PILEUP_WINDOW * 8];
    __shared__ unsigned long long wkeys[PILEUP_WINDOW]; // the window's panel keys: every search of an in-window run stays in LDS
    __shared__ long long win_base;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long long first = (long long)blockIdx.x * PILEUP_READS;
    for (int i = threadIdx.x; i < PILEUP_WINDOW * 8; i += 256) win[i] = 0;
    if (threadIdx.x == 0) { // the window starts at the first panel position at or behind the start of the workgroup's first placed read
        long long wb = 0;
        for (long long rd = first; rd < n_reads && rd < first + PILEUP_READS; ++rd) {
            const unsigned char *r = bam + rec_off[rd] + 4;
            const int ref_id = (int)ld_u32_unaligned(r), pos = (int)ld_u32_unaligned(r + 4);
            if (ref_id < 0 || pos < 0) continue;
            const unsigned long long k0 = ((unsigned long long)(unsigned)ref_id << 32) | (unsigned long long)((long long)pos + 1);
            long long lo = 0, hi = P;
            while (lo < hi) {
                const long long mid = (lo + hi) >> 1;
                if (keys[mid] < k0) lo = mid + 1;
                else hi = mid;
            }
            wb = lo;
            break;
        }
        win_base = wb;
    }
    __syncthreads();
    const long long wb = win_base;
    const int nw = (int)(P - wb < PILEUP_WINDOW ? P - wb : PILEUP_WINDOW);
    for (int i = threadIdx.x; i < nw; i += 256) wkeys[i] = keys[wb + i];
    __syncthreads();
    unsigned long long kept = 0, added = 0;
    unsigned long long prev_k0 = ~0ull; // the reads of an amplicon start at the same place: the last run's search is usually this run's
    int prev_lo = 0;
    for (long long read = first + wave; read < n_reads && read < first + PILEUP_READS; read += 4) {
        const unsigned char *r = bam + rec_off[read] + 4; // past block_size
        const int ref_id = (int)ld_u32_unaligned(r), pos = (int)ld_u32_unaligned(r + 4);
        const unsigned bin_mq_nl = ld_u32_unaligned(r + 8), flag_nc = ld_u32_unaligned(r + 12);
        const int l_read_name = (int)(bin_mq_nl & 0xffu), mapq = (int)((bin_mq_nl >> 8) & 0xffu);
        const int n_cigar = (int)(flag_nc & 0xffffu);
        const unsigned flag = flag_nc >> 16;
        if (ref_id < 0 || pos < 0 || (flag & (0x4u | 0x100u | 0x200u | 0x400u)) != 0 || mapq < mrq) continue;
        ++kept;
        const int l_seq = (int)ld_u32_unaligned(r + 16);
        const unsigned char *cig = r + 32 + l_read_name;
        const unsigned char *seq = cig + 4 * (size_t)n_cigar;
        const unsigned char *qual = seq + (l_seq + 1) / 2;
        const int rev = (int)((flag >> 4) & 1u);
        long long refpos = pos; // 0-based
        int qpos = 0;
        for (int c = 0; c < n_cigar; ++c) {
            const unsigned op_len = ld_u32_unaligned(cig + 4 * (size_t)c);
            const unsigned op = op_len & 15u;
            const int len = (int)(op_len >> 4);
            if (op == 0 || op == 7 || op == 8) { // M, =, X
                const unsigned long long k0 = ((unsigned long long)(unsigned)ref_id << 32) | (unsigned long long)(refpos + 1);
                // the run lies inside the window's key range: everything below happens in LDS
                const bool inwin = nw > 0 && k0 >= wkeys[0] && k0 + (unsigned long long)len - 1 <= wkeys[nw - 1];
                if (inwin) {
                    int lo;
                    if (k0 == prev_k0) {
                        lo = prev_lo;
                    } else { // first window key >= k0 (wave-uniform)
                        int l = 0, h = nw;
                        while (l < h) {
                            const int mid = (l + h) >> 1;
                            if (wkeys[mid] < k0) l = mid + 1;
                            else h = mid;
                        }
                        lo = l;
                        prev_k0 = k0;
                        prev_lo = l;
                    }
                    const int wend = lo + len < nw ? lo + len : nw; // the keys of this run are among the next `len`
                    for (int j = lane; j < len; j += 64) {
                        const unsigned long long key = k0 + (unsigned long long)j;
                        int a = lo + j < wend ? lo + j : wend - 1; // no gap in the panel along this run: the key sits at lo + j
                        if (wkeys[a] != key) {
                            int b = wend;
                            a = lo;
                            while (a < b) {
                                const int mid = (a + b) >> 1;
                                if (wkeys[mid] < key) a = mid + 1;
                                else b = mid;
                            }
                        }
                        if (a < wend && wkeys[a] == key) {
                            const int q = qpos + j;
                            const unsigned nib = (seq[q >> 1] >> ((~q & 1) * 4)) & 15u;
                            const int b4 = nib == 1u ? 0 : (nib == 2u ? 1 : (nib == 4u ? 2 : (nib == 8u ? 3 : -1)));
                            if (b4 >= 0 && (int)qual[q] >= mbq) {
                                atomicAdd(&win[a * 8 + b4], 1);
                                if (rev) atomicAdd(&win[a * 8 + 4 + b4], 1);
                                ++added;
                            }
                        }
                    }
                } else {
                    // outside the window (unsorted file, a run that leaves the window, a position before it): global search and counters
                    long long lo = 0, hi = P;
                    while (lo < hi) {
                        const long long mid = (lo + hi) >> 1;
                        if (keys[mid] < k0) lo = mid + 1;
                        else hi = mid;
                    }
                    const long long wend = lo + len < P ? lo + len : P;
                    if (lo < wend && keys[lo] < k0 + (unsigned long long)len) {
                        for (int j = lane; j < len; j += 64) {
                            const unsigned long long key = k0 + (unsigned long long)j;
                            long long a = lo, b = wend;
                            while (a < b) {
                                const long long mid = (a + b) >> 1;
                                if (keys[mid] < key) a = mid + 1;
                                else b = mid;
                            }
                            if (a < wend && keys[a] == key) {
                                const int q = qpos + j;
                                const unsigned nib = (seq[q >> 1] >> ((~q & 1) * 4)) & 15u;
                                const int b4 = nib == 1u ? 0 : (nib == 2u ? 1 : (nib == 4u ? 2 : (nib == 8u ? 3 : -1)));
                                if (b4 >= 0 && (int)qual[q] >= mbq) {
                                    atomicAdd(&counts[a * 8 + b4], 1);
                                    if (rev) atomicAdd(&counts[a * 8 + 4 + b4], 1);
                                    ++added;
                                }
                            }
                        }
                    }
                }
                refpos += len;
                qpos += len;
            } else if (op == 1 || op == 4) { // I, S
                qpos += len;
            } else if (op == 2 || op == 3) { // D, N
                refpos += len;
            } // H, P: neither
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PILEUP_WINDOW * 8; i += 256) {
        const int v = win[i];
        if (v != 0 && wb + (i >> 3) < P) atomicAdd(&counts[(wb + (i >> 3)) * 8 + (i & 7)], v);
    }
    if (stats) {
        if (lane == 0 && kept) atomicAdd(&stats[0], kept); // reads kept
        if (added) atomicAdd(&stats[1], added);            // bases counted
    }
}

// ---------------------------------------------------------------------------
// The staged form (default): the wave-per-read kernel above spends its time on the dependent loads of ONE record at a time
// per wave (header -> CIGAR -> bases: 2.5 ms for 2 M reads, 185 GB/s).  Here a workgroup first copies the bytes of its
// PILEUP_READS consecutive records -- they are contiguous in the stream -- into LDS with coalesced 16-byte loads, and then every
// THREAD walks one read out of LDS: 64 records in flight per wave instead of one, no global load in the walk at all.  The
// reads of an amplicon start at the same place, so lane l starts its walk of a match run l * PILEUP_ROT bases in (wrapping
// around): at any step the lanes of a wave then add to different counters of the LDS window instead of all to the same one.
// A group whose records do not fit the stage (very long reads) falls back to the wave-per-read walk from global memory.
// ---------------------------------------------------------------------------
constexpr int PILEUP_STAGE = 56 * 1024; // bytes of records a workgroup stages
constexpr int PILEUP_SWINDOW = 384;     // panel positions of the staged kernel's LDS window
constexpr int PILEUP_ROT = 3;

__device__ __forceinline__ unsigned lds_u32(const unsigned char *p) // unaligned little-endian dword out of LDS
{
    const unsigned *q = (const unsigned *)((uintptr_t)p & ~(uintptr_t)3);
    const unsigned sh = (unsigned)((uintptr_t)p & 3u);
    const unsigned lo = q[0], hi = q[1];
    return sh ? __builtin_amdgcn_alignbyte(hi, lo, sh) : lo;
}

__global__ __launch_bounds__(256) void pileup_count_staged_kernel(const unsigned char *__restrict__ bam, const unsigned long long *__restrict__ rec_off,
                                                                  const long long n_reads, const unsigned long long *__restrict__ keys, const long long P,
                                                                  const int mbq, const int mrq, int *__restrict__ counts,
                                                                  unsigned long long *__restrict__ stats)
{
    __shared__ __attribute__((aligned(16))) unsigned char stage[PILEUP_STAGE + 32];
    __shared__ int win[PILEUP_SWINDOW * 8];
    __shared__ unsigned long long wkeys[PILEUP_SWINDOW];
    __shared__ long long s_wb;
    __shared__ unsigned long long s_span[2]; // first byte (16-byte aligned down) and end of the group's records in the stream
    const long long first = (long long)blockIdx.x * PILEUP_READS;
    const int n_here = (int)(n_reads - first < PILEUP_READS ? n_reads - first : PILEUP_READS);
    for (int i = threadIdx.x; i < PILEUP_SWINDOW * 8; i += 256) win[i] = 0;
    if (threadIdx.x == 0) {
        const unsigned long long b0 = rec_off[first], last = rec_off[first + n_here - 1];
        s_span[0] = b0 & ~15ull;
        s_span[1] = last + 4ull + (unsigned long long)ld_u32_unaligned(bam + last);
        long long wb = 0; // the window starts at the first panel position at or behind the start of the group's first placed read
        for (int k = 0; k < n_here; ++k) {
            const unsigned char *r = bam + rec_off[first + k] + 4;
            const int ref_id = (int)ld_u32_unaligned(r), pos = (int)ld_u32_unaligned(r + 4);
            if (ref_id < 0 || pos < 0) continue;
            const unsigned long long k0 = ((unsigned long long)(unsigned)ref_id << 32) | (unsigned long long)((long long)pos + 1);
            long long lo = 0, hi = P;
            while (lo < hi) {
                const long long mid = (lo + hi) >> 1;
                if (keys[mid] < k0) lo = mid + 1;
                else hi = mid;
            }
            wb = lo;
            break;
        }
        s_wb = wb;
    }
    __syncthreads();
    const long long wb = s_wb;
    const unsigned long long g0 = s_span[0], g1 = s_span[1];
    const bool staged = g1 - g0 <= (unsigned long long)PILEUP_STAGE; // workgroup-uniform
    const int nw = (int)(P - wb < PILEUP_SWINDOW ? P - wb : PILEUP_SWINDOW);
    for (int i = threadIdx.x; i < nw; i += 256) wkeys[i] = keys[wb + i];
    if (staged) { // coalesced copy of the group's bytes (the host pads the buffer so that whole 16-byte pieces can be read)
        const uint4 *src = (const uint4 *)(bam + g0);
        uint4 *dst = (uint4 *)stage;
        const int n16 = (int)((g1 - g0 + 15) >> 4);
        for (int i = threadIdx.x; i < n16; i += 256) dst[i] = src[i];
    }
    __syncthreads();
    unsigned long long kept = 0, added = 0;
    if (staged) {
        const int t = threadIdx.x;
        if (t < n_here) {
            const unsigned char *r = stage + (rec_off[first + t] - g0) + 4; // past block_size
            const int ref_id = (int)lds_u32(r), pos = (int)lds_u32(r + 4);
            const unsigned bin_mq_nl = lds_u32(r + 8), flag_nc = lds_u32(r + 12);
            const int l_read_name = (int)(bin_mq_nl & 0xffu), mapq = (int)((bin_mq_nl >> 8) & 0xffu);
            const int n_cigar = (int)(flag_nc & 0xffffu);
            const unsigned flag = flag_nc >> 16;
            if (!(ref_id < 0 || pos < 0 || (flag & (0x4u | 0x100u | 0x200u | 0x400u)) != 0 || mapq < mrq)) {
                kept = 1;
                const int l_seq = (int)lds_u32(r + 16);
                const unsigned char *cig = r + 32 + l_read_name;
                const unsigned char *seq = cig + 4 * (size_t)n_cigar;
                const unsigned char *qual = seq + (l_seq + 1) / 2;
                const int rev = (int)((flag >> 4) & 1u);
                const int lane = t & 63;
                long long refpos = pos;
                int qpos = 0;
                for (int c = 0; c < n_cigar; ++c) {
                    const unsigned op_len = lds_u32(cig + 4 * (size_t)c);
                    const unsigned op = op_len & 15u;
                    const int len = (int)(op_len >> 4);
                    if (op == 0 || op == 7 || op == 8) { // M, =, X
                        const unsigned long long k0 = ((unsigned long long)(unsigned)ref_id << 32) | (unsigned long long)(refpos + 1);
                        const bool inwin = len > 0 && nw > 0 && k0 >= wkeys[0] && k0 + (unsigned long long)len - 1 <= wkeys[nw - 1];
                        int lo = 0;
                        long long glo = 0, gend = 0;
                        if (inwin) { // first window key >= k0
                            int l = 0, h = nw;
                            while (l < h) {
                                const int mid = (l + h) >> 1;
                                if (wkeys[mid] < k0) l = mid + 1;
                                else h = mid;
                            }
                            lo = l;
                        } else if (len > 0) { // outside the window: global keys, global counters
                            long long l = 0, h = P;
                            while (l < h) {
                                const long long mid = (l + h) >> 1;
                                if (keys[mid] < k0) l = mid + 1;
                                else h = mid;
                            }
                            glo = l;
                            gend = l + len < P ? l + len : P;
                        }
                        const int wend = lo + len < nw ? lo + len : nw;
                        int j = len > 0 ? (lane * PILEUP_ROT) % len : 0; // rotated start: the lanes of a wave hit different counters
                        for (int step = 0; step < len; ++step, j = j + 1 == len ? 0 : j + 1) {
                            const unsigned long long key = k0 + (unsigned long long)j;
                            long long a = -1; // index of the key among the panel positions, -1: not a panel position
                            if (inwin) {
                                int x = lo + j < wend ? lo + j : wend - 1;
                                if (wkeys[x] != key) {
                                    int b = wend;
                                    x = lo;
                                    while (x < b) {
                                        const int mid = (x + b) >> 1;
                                        if (wkeys[mid] < key) x = mid + 1;
                                        else b = mid;
                                    }
                                }
                                if (x < wend && wkeys[x] == key) a = x;
                            } else if (glo < gend) {
                                long long x = glo, b = gend;
                                while (x < b) {
                                    const long long mid = (x + b) >> 1;
                                    if (keys[mid] < key) x = mid + 1;
                                    else b = mid;
                                }
                                if (x < gend && keys[x] == key) a = x;
                            }
                            if (a < 0) continue;
                            const int q = qpos + j;
                            const unsigned nib = (seq[q >> 1] >> ((~q & 1) * 4)) & 15u;
                            const int b4 = nib == 1u ? 0 : (nib == 2u ? 1 : (nib == 4u ? 2 : (nib == 8u ? 3 : -1)));
                            if (b4 < 0 || (int)qual[q] < mbq) continue;
                            if (inwin) {
                                atomicAdd(&win[a * 8 + b4], 1);
                                if (rev) atomicAdd(&win[a * 8 + 4 + b4], 1);
                            } else {
                                atomicAdd(&counts[a * 8 + b4], 1);
                                if (rev) atomicAdd(&counts[a * 8 + 4 + b4], 1);
                            }
                            ++added;
                        }
                        refpos += len;
                        qpos += len;
                    } else if (op == 1 || op == 4) { // I, S
                        qpos += len;
                    } else if (op == 2 || op == 3) { // D, N
                        refpos += len;
                    } // H, P: neither
                }
            }
        }
    } else {
        // records too long for the stage: one wave per read at a time, straight from global memory, global counters
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (long long read = first + wave; read < first + n_here; read += 4) {
            const unsigned char *r = bam + rec_off[read] + 4;
            const int ref_id = (int)ld_u32_unaligned(r), pos = (int)ld_u32_unaligned(r + 4);
            const unsigned bin_mq_nl = ld_u32_unaligned(r + 8), flag_nc = ld_u32_unaligned(r + 12);
            const int l_read_name = (int)(bin_mq_nl & 0xffu), mapq = (int)((bin_mq_nl >> 8) & 0xffu);
            const int n_cigar = (int)(flag_nc & 0xffffu);
            const unsigned flag = flag_nc >> 16;
            if (ref_id < 0 || pos < 0 || (flag & (0x4u | 0x100u | 0x200u | 0x400u)) != 0 || mapq < mrq) continue;
            if (lane == 0) ++kept;
            const int l_seq = (int)ld_u32_unaligned(r + 16);
            const unsigned char *cig = r + 32 + l_read_name;
            const unsigned char *seq = cig + 4 * (size_t)n_cigar;
            const unsigned char *qual = seq + (l_seq + 1) / 2;
            const int rev = (int)((flag >> 4) & 1u);
            long long refpos = pos;
            int qpos = 0;
            for (int c = 0; c < n_cigar; ++c) {
                const unsigned op_len = ld_u32_unaligned(cig + 4 * (size_t)c);
                const unsigned op = op_len & 15u;
                const int len = (int)(op_len >> 4);
                if (op == 0 || op == 7 || op == 8) {
                    const unsigned long long k0 = ((unsigned long long)(unsigned)ref_id << 32) | (unsigned long long)(refpos + 1);
                    long long lo = 0, hi = P;
                    while (lo < hi) {
                        const long long mid = (lo + hi) >> 1;
                        if (keys[mid] < k0) lo = mid + 1;
                        else hi = mid;
                    }
                    const long long wend = lo + len < P ? lo + len : P;
                    if (lo < wend && keys[lo] < k0 + (unsigned long long)len) {
                        for (int j = lane; j < len; j += 64) {
                            const unsigned long long key = k0 + (unsigned long long)j;
                            long long a = lo, b = wend;
                            while (a < b) {
                                const long long mid = (a + b) >> 1;
                                if (keys[mid] < key) a = mid + 1;
                                else b = mid;
                            }
                            if (a < wend && keys[a] == key) {
                                const int q = qpos + j;
                                const unsigned nib = (seq[q >> 1] >> ((~q & 1) * 4)) & 15u;
                                const int b4 = nib == 1u ? 0 : (nib == 2u ? 1 : (nib == 4u ? 2 : (nib == 8u ? 3 : -1)));
                                if (b4 >= 0 && (int)qual[q] >= mbq) {
                                    atomicAdd(&counts[a * 8 + b4], 1);
                                    if (rev) atomicAdd(&counts[a * 8 + 4 + b4], 1);
                                    ++added;
                                }
                            }
                        }
                    }
                    refpos += len;
                    qpos += len;
                } else if (op == 1 || op == 4) {
                    qpos += len;
                } else if (op == 2 || op == 3) {
                    refpos += len;
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PILEUP_SWINDOW * 8; i += 256) {
        const int v = win[i];
        if (v != 0 && wb + (i >> 3) < P) atomicAdd(&counts[(wb + (i >> 3)) * 8 + (i & 7)], v);
    }
    if (stats) {
        if (kept) atomicAdd(&stats[0], kept);
        if (added) atomicAdd(&stats[1], added);
    }
}

extern "C" int ampli_pileup_count(ampli_ctx *ctx, const uint8_t *d_bam, const uint64_t *d_rec_off, int64_t n_reads, const uint64_t *d_keys, int64_t P,
                                  int32_t mbq, int32_t mrq, int32_t *d_counts, uint64_t *d_stats)
{
    if (!ctx) return AMPLI_E_INVALID;
    if (!d_bam || !d_rec_off || n_reads < 0 || !d_keys || P <= 0 || !d_counts) return fail(ctx, AMPLI_E_INVALID, "pileup_count: bad argument");
    if (n_reads == 0) return AMPLI_OK;
    if ((n_reads + PILEUP_READS - 1) / PILEUP_READS > 0x7fffffffll) return fail(ctx, AMPLI_E_RANGE, "pileup_count: too many reads in one call; split the batch");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // AMPLI_PILEUP_WAVE_PER_READ (build-time) keeps the first form for comparison (tools/pileup_bench.py)
#ifdef AMPLI_PILEUP_WAVE_PER_READ
    const auto kernel = pileup_count_kernel;
#else
    const auto kernel = pileup_count_staged_kernel;
#endif
    hipLaunchKernelGGL(kernel, dim3((unsigned)((n_reads + PILEUP_READS - 1) / PILEUP_READS)), dim3(256), 0, main_stream(ctx), (const unsigned char *)d_bam,
                       (const unsigned long long *)d_rec_off, (long long)n_reads, (const unsigned long long *)d_keys, (long long)P, (int)mbq, (int)mrq,
                       d_counts, (unsigned long long *)d_stats);
    return check_launch(ctx, "pileup_count_kernel");
}
