// HIP kernels (gfx950 / CDNA4, wave64) of the fragment-feature engine.
//
// All kernels are integer compare/count/scan work bound by HBM bandwidth: no
// MFMA.  Design notes live in DESIGN.md; the short version:
//   * fragments of a contig are a start-sorted SoA in HBM plus a coarse
//     position index (first fragment per 512-bp bin), so the candidate
//     fragments of a window / WPS tile are one contiguous index range;
//   * window features (coverage, DELFI, length histogram) are computed
//     window-centrically: small candidate ranges one wave per window, large
//     ones cut into 4096-fragment chunks that a fixed grid walks in order;
//   * WPS (and the cleavage profile) build a per-tile difference array in LDS
//     with ds_add, scan it on DPP and stream the scores out with non-temporal
//     16-byte stores, 1 KB contiguous per store instruction.
#include "ftk_kernels.h"

#include <cstdlib>

#include <algorithm>

namespace ftk {

// ---------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------
__device__ __forceinline__ int wave_reduce_add(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// ---------------------------------------------------------------------------
// load-time kernels
// ---------------------------------------------------------------------------
// Validation summary of a freshly loaded contig (sortedness, length / coordinate extremes).  Reduced in the wave and
// then in the block, ONE set of five atomics per block: device-scope atomics on one cache line retire at ~9 ns each,
// and with a set per wave of 2048 x 256 threads (41 k sets) the kernel took 380 us whatever the contig's size - half
// the GPU time of a small end-to-end run.  Blocks of 256 threads (round 4; 1024 before): a contig is loaded while the
// decoder's inflate waves hold five of a SIMD's wave slots and 410 of its 512 VGPRs on every CU - a 1 024-thread block
// (four waves per SIMD at 68 VGPRs) could only start where inflate waves had retired and the kernel took 1.1-1.4 ms
// beside them against 22 us alone (profiles/r4_a_genome_leg_kernel_stats.txt); one wave per SIMD fits at once.
constexpr int kStatsThreads = 256;
__global__ __launch_bounds__(kStatsThreads) void stats_kernel(const int32_t* start, const int32_t* end, int n, FragStats* st) {
    __shared__ int red[5][kStatsThreads / 64];
    int unsorted = 0, max_len = INT32_MIN, min_len = INT32_MAX, max_end = INT32_MIN, min_start = INT32_MAX;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        int s = start[i], e = end[i];
        if (i > 0 && start[i - 1] > s) unsorted = 1;
        int len = e - s;
        max_len = max(max_len, len);
        min_len = min(min_len, len);
        max_end = max(max_end, e);
        min_start = min(min_start, s);
    }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) {
        unsorted |= __shfl_xor(unsorted, d, 64);
        max_len = max(max_len, __shfl_xor(max_len, d, 64));
        min_len = min(min_len, __shfl_xor(min_len, d, 64));
        max_end = max(max_end, __shfl_xor(max_end, d, 64));
        min_start = min(min_start, __shfl_xor(min_start, d, 64));
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][wv] = unsorted; red[1][wv] = max_len; red[2][wv] = min_len; red[3][wv] = max_end; red[4][wv] = min_start;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < kStatsThreads / 64; ++k) {
            unsorted |= red[0][k];
            max_len = max(max_len, red[1][k]);
            min_len = min(min_len, red[2][k]);
            max_end = max(max_end, red[3][k]);
            min_start = min(min_start, red[4][k]);
        }
        if (unsorted) atomicOr(&st->unsorted, 1);
        atomicMax(&st->max_len, max_len);
        atomicMin(&st->min_len, min_len);
        atomicMax(&st->max_end, max_end);
        atomicMin(&st->min_start, min_start);
    }
}

__global__ void bin_index_kernel(const int32_t* start, int n, int n_bins, int32_t* idx) {
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k > n_bins) return;
    if (k == n_bins) { idx[k] = n; return; }
    long long p = (long long)k << kBinShift;
    int lo = 0, hi = n;
    while (lo < hi) {
        int m = (lo + hi) >> 1;
        if (start[m] < p) lo = m + 1; else hi = m;
    }
    idx[k] = lo;
}

// Does every fragment hold its read1 span (start <= r1_start < r1_end <= end)?  *bad is set when one does not
// (ContigView::r1_inside; io/alignment.py:254-261 builds the fragment from read1's start or end and TLEN, so the
// span can only stick out when TLEN is shorter than the read's own alignment).
__global__ __launch_bounds__(256) void r1_inside_kernel(const int32_t* start, const int32_t* end, const int32_t* r1s,
                                                        const int32_t* r1e, int n, int* bad) {
    bool b = false;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int rs = r1s[i], re = r1e[i];
        b |= !(start[i] <= rs && rs < re && re <= end[i]);
    }
    if (__ballot(b) && (threadIdx.x & 63) == 0) atomicOr(bad, 1);
}

// ---------------------------------------------------------------------------
// window planning: candidate range per window, chunk counts, chunk offsets
// ---------------------------------------------------------------------------
// A fragment can only pass a window [ws, we) (either policy, and the tabix
// overlap query itself) if fs < we and fe > ws - 1, hence
// ws - lmax <= fs < we with lmax = longest admissible fragment.  The range is
// taken straight from the 512-bp index (conservative by < 1 bin on each side;
// the predicate decides), so planning costs two index reads per window and no
// dependent bisection.
__device__ __forceinline__ void window_candidates(const ContigView& cv, int s, int e, int lmax, int small_max, int& lo,
                                                  int& hi, uint32_t& nchunks) {
    const long long ql = (long long)s - lmax;
    if (s == INT32_MIN || ql <= 0) lo = 0;
    else { const long long k = ql >> kBinShift; lo = k >= cv.n_bins ? cv.n : cv.bin_idx[k]; }
    if (e == INT32_MAX) hi = cv.n;
    else if (e <= 0) hi = 0;
    else { const long long k = (long long)e >> kBinShift; hi = k >= cv.n_bins ? cv.n : cv.bin_idx[k + 1]; }
    lo &= ~3;  // chunks then start on 16-byte boundaries of the columns
    if (hi < lo || e < s) hi = lo;
    const int cnt = hi - lo;
    nchunks = (cnt <= small_max) ? 0u : (uint32_t)((cnt + kChunk - 1) / kChunk);
}

constexpr int kPlanSingleBlockMax = 16384;  // windows one plan_kernel block handles (16 per thread)

struct ZeroList {  // outputs the chunked path accumulates into with atomics
    int64_t* p[4];
};

__global__ void bounds_kernel(ContigView cv, const int32_t* ws, const int32_t* we, int n_win, int lmax,
                              int small_max, int32_t* cand_lo, int32_t* cand_hi, uint32_t* nchunks, ZeroList z) {
    int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= n_win) return;
    int lo, hi;
    uint32_t nc;
    window_candidates(cv, ws[w], we[w], lmax, small_max, lo, hi, nc);
    cand_lo[w] = lo;
    cand_hi[w] = hi;
    nchunks[w] = nc;
    // the chunked path accumulates with atomics: start from zero.  Without a wave-per-window pass
    // (small_max < 0) nobody else writes the rows of windows that have no candidates at all.
    if (nc || small_max < 0) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (z.p[k]) z.p[k][w] = 0;
    }
}

// Exclusive scan of nchunks[0..n) into off[0..n]; one 1024-thread block.
__global__ __launch_bounds__(1024) void scan_kernel(const uint32_t* nchunks, int n, uint32_t* off) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        int i = base + tid;
        uint32_t v = (i < n) ? nchunks[i] : 0u;
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wave_tot[wv] = x;
        __syncthreads();
        uint32_t pre = carry_s;
        for (int j = 0; j < wv; ++j) pre += wave_tot[j];
        if (i < n) off[i] = pre + x - v;
        __syncthreads();
        if (tid == 1023) carry_s = pre + x;
        __syncthreads();
    }
    if (tid == 0) off[n] = carry_s;
}

// Candidate ranges + chunk offsets in ONE single-block launch (n_win up to a
// few thousand per contig is the common case: saves two dependent launches).
__global__ __launch_bounds__(1024) void plan_kernel(ContigView cv, const int32_t* ws, const int32_t* we, int n_win,
                                                    int lmax, int small_max, int32_t* cand_lo, int32_t* cand_hi,
                                                    uint32_t* nchunks, uint32_t* off, ZeroList z) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    // phase 1: every window's candidate range (up to 16 independent index lookups per thread in flight)
    constexpr int R = kPlanSingleBlockMax / 1024;
    uint32_t vv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int w = r * 1024 + tid;
        vv[r] = 0;
        if (w < n_win) {
            int lo, hi;
            window_candidates(cv, ws[w], we[w], lmax, small_max, lo, hi, vv[r]);
            cand_lo[w] = lo;
            cand_hi[w] = hi;
            nchunks[w] = vv[r];
            if (vv[r] || small_max < 0) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (z.p[k]) z.p[k][w] = 0;
            }
        }
    }
    __syncthreads();
    // phase 2: exclusive scan of the chunk counts
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (r * 1024 >= n_win) break;
        const int w = r * 1024 + tid;
        const uint32_t v = vv[r];
        uint32_t x = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            uint32_t y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        if (lane == 63) wave_tot[wv] = x;
        __syncthreads();
        uint32_t pre = carry_s;
        for (int j = 0; j < wv; ++j) pre += wave_tot[j];
        if (w < n_win) off[w] = pre + x - v;
        __syncthreads();
        if (tid == 1023) carry_s = pre + x;
        __syncthreads();
    }
    if (tid == 0) off[n_win] = carry_s;
}

// ---------------------------------------------------------------------------
// predicates
// ---------------------------------------------------------------------------
// utils/_frag_generator.py:117-130 + io/alignment.py:291 (mapq) + the index
// query that produced the stream (io/alignment.py:270-279 tabix overlap, or
// :245 read1 overlap for BAM).  Returns 1 when the fragment counts.
struct WinPred {
    int mapq_min, min_len, max_len, policy, bam;
    // Branch-free on purpose: the comparisons become v_cmp + scalar mask logic; a short-circuit chain
    // costs an exec-mask save / branch per term (the fused pass was ALU-bound on exactly that).
    template <bool BAM>
    __device__ __forceinline__ int test(const ContigView& cv, int i, int fs, int fe, int q, int ws, int we) const {
        const int len = fe - fs;
        bool ok = (q >= mapq_min) & (len >= min_len) & (len <= max_len);
        const bool overlap = (fs < we) & (fe > ws);
        if (BAM) {
            const int rs = cv.r1_start[i], re = cv.r1_end[i];
            ok &= (rs < we) & (re > ws);
        } else {
            ok &= overlap;
        }
        const int mid = (int)(((unsigned)fs + (unsigned)fe) >> 1);  // coordinates < 2^30
        const bool mid_in = (mid >= ws) & (mid < we);
        ok &= (policy == FTK_POLICY_MIDPOINT) ? mid_in : (policy == FTK_POLICY_ANY) ? overlap : true;  // FETCH: the query alone
        return ok;
    }
    __device__ __forceinline__ int operator()(const ContigView& cv, int i, int fs, int fe, int q, int ws, int we,
                                              int /*w*/) const {
        return bam ? test<true>(cv, i, fs, fe, q, ws, we) : test<false>(cv, i, fs, fe, q, ws, we);
    }
};

// frag/_delfi.py:443-472.  Returns 0 (skip), 1 (short) or 2 (long).  [o0, o1) is
// the window's slice of the blacklist CSR (hoisted per window by the caller).
// ContigGaps.in_tcmere (genome/gaps.py:217-237) is pre-reduced on the host to two
// intervals: the centromere [cen0, cen1) and the telomere condition
// all_i(stop > t0_i and start < t1_i) == stop > max_i t0_i and start < min_i t1_i
// (tel0 = INT32_MAX when there are no telomeres / no gap annotation at all).
struct DelfiPred {
    int mapq_min;
    int cen0, cen1, tel0, tel1;
    const int32_t* bl_off;  // n_win + 1 offsets into bl_r0 / bl_pm (NULL: no blacklist)
    const int32_t* bl_r0;   // region starts (sorted) of the regions fully inside each window
    const int32_t* bl_pm;   // running maximum of the region stops inside each window
    template <bool BAM>
    __device__ __forceinline__ int test(const ContigView& cv, int i, int fs, int fe, int q, int ws, int we, int o0,
                                        int o1, bool valid = true) const {
        const int len = fe - fs;
        const int mid = (int)(((unsigned)fs + (unsigned)fe) >> 1);  // coordinates < 2^30
        bool ok = valid & (q >= mapq_min) & ((unsigned)(len - 100) <= 120u) & (mid >= ws) & (mid < we);
        if (BAM) {
            const int rs = cv.r1_start[i], re = cv.r1_end[i];
            ok &= (rs < we) & (re > ws);
        } else {
            ok &= (fs < we) & (fe > ws);
        }
        ok &= !((fe > cen0) & (fs < cen1)) & !((fe > tel0) & (fs < tel1));
        if (o1 > o0) {  // (uniform per window) frag/_delfi.py:455-462: blacklisted iff max{r1 : r0 <= fs} > fe
            if (ok) {
                int lo = o0, hi = o1;  // upper bound: first region with r0 > fs
                while (lo < hi) {
                    const int m = (lo + hi) >> 1;
                    if (bl_r0[m] <= fs) lo = m + 1; else hi = m;
                }
                if (lo > o0 && bl_pm[lo - 1] > fe) ok = false;
            }
        }
        return ok ? ((len >= 151) ? 2 : 1) : 0;
    }
};

// CH of the window-feature kernels: 0 no coverage filter, 1 coverage / histogram, 2 motif pass (any image, any k),
// kMotifWord the motif pass on a 2bit image with k <= 13.
constexpr int kMotifWord = 3;

// k-mer starting at base `lo` of the reference image as a base-4 number in ACGT order
// (gen_kmers order, utils/utils.py:388-410), or -1 when it holds anything but A/C/G/T
// (upper- or lower-case: io/reference.py:171 upper-cases).  revcomp: the reverse complement's code.
__device__ __forceinline__ int kmer_code(const MotifParams& M, int lo, bool revcomp) {
    int code = 0;
    if (M.kind == FTK_REF_2BIT) {
        // N blocks (sorted, disjoint): first block ending after lo
        int a = 0, b = M.n_nblk;
        while (a < b) {
            const int m = (a + b) >> 1;
            if (M.nblk_end[m] <= lo) a = m + 1; else b = m;
        }
        if (a < M.n_nblk && M.nblk_start[a] < lo + M.k) return -1;
        if (M.k <= 13) {
            // bases lo .. lo + k - 1 lie in at most four consecutive bytes: ONE (unaligned) 4-byte load instead of a
            // byte load per base (the image's block is at least 32 bytes longer than the image), the sixteen 2-bit
            // groups translated at once, the k-mer cut out with a shift and a mask
            typedef uint32_t __attribute__((aligned(1))) u32u;
            const uint32_t be = __builtin_bswap32(*reinterpret_cast<const u32u*>(M.img + (lo >> 2)));  // first base on top
            const uint32_t v1 = (be >> 1) & 0x55555555u, v0 = be & 0x55555555u;    // T=00 C=01 A=10 G=11
            const uint32_t acgt = ((~(v1 ^ v0) & 0x55555555u) << 1) | (~v1 & 0x55555555u);  // -> A=00 C=01 G=10 T=11
            const uint32_t mask = (1u << (2 * M.k)) - 1u;
            uint32_t x = (acgt >> (32 - 2 * ((lo & 3) + M.k))) & mask;  // base 0 most significant
            if (revcomp) {  // base j complemented at bits 2j: the 2-bit groups in reverse order, each XOR 3
                uint32_t y = __brev(x) >> (32 - 2 * M.k);
                y = ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u);
                x = y ^ mask;
            }
            return (int)x;
        }
        for (int j = 0; j < M.k; ++j) {
            const int p = lo + j;
            const int v = (M.img[p >> 2] >> (6 - 2 * (p & 3))) & 3;   // T=0 C=1 A=2 G=3
            const int base = (0x87 >> (2 * v)) & 3;                   // -> A=0 C=1 G=2 T=3 ({3,1,0,2} packed)
            code = revcomp ? (code | ((3 - base) << (2 * j))) : (code * 4 + base);
        }
        return code;
    }
    int row = lo / M.line_bases, col = lo - row * M.line_bases;
    long long off = (long long)row * M.line_width + col;
    for (int j = 0; j < M.k; ++j) {
        const int ch = M.img[off] & 0xDF;
        int base;
        if (ch == 'A') base = 0;
        else if (ch == 'C') base = 1;
        else if (ch == 'G') base = 2;
        else if (ch == 'T') base = 3;
        else return -1;
        code = revcomp ? (code | ((3 - base) << (2 * j))) : (code * 4 + base);
        ++off;
        if (++col == M.line_bases) { col = 0; off += M.line_width - M.line_bases; }
    }
    return code;
}

// ---- the motif pass with its reference loads batched ---------------------------------------------------------
// A fragment of the motif pass costs two gathers from the reference image.  Done one element at a time (test, N-block
// search, load, atomic) every element is a chain of dependent loads and the pass is bound by their latency (162 us for
// the 24 M fragments of a chr2-sized contig, 16 elements per thread in a row).  Here the N elements a thread holds are
// tested first, ALL their 2 N words requested (an element that contributes nothing loads word 0), and only then turned
// into codes and LDS atomics.  The N-block search is cut down per window: [o0, o1) are the blocks within reach of the
// window's fragments (motif_window), none at all for most windows.

// Positions a fetched fragment's k-mers can start at: (g_lo, g_hi).  fs < we, fe > ws, fe - fs <= max_len.
__device__ __forceinline__ void motif_reach(const MotifParams& M, const ContigView& cv, int ws, int we, int& g_lo, int& g_hi) {
    const long long r = (long long)cv.max_len + abs(M.f_off) + abs(M.r_off) + 1;
    const long long lo = (long long)ws - r, hi = (long long)we + r + M.k;
    g_lo = (int)max(lo, (long long)INT32_MIN / 2);
    g_hi = (int)min(hi, (long long)INT32_MAX / 2);
}

// N blocks [o0, o1) that a k-mer starting inside the window's reach can touch.  The blocks are sorted and disjoint:
// o0 = blocks ending at or before g_lo, o1 = blocks starting before g_hi, counted 64 at a time by the wave.
__device__ __forceinline__ void motif_window(const MotifParams& M, const ContigView& cv, int ws, int we, int& o0, int& o1) {
    o0 = o1 = 0;
    if (M.kind != FTK_REF_2BIT || M.n_nblk == 0) return;
    int g_lo, g_hi;
    motif_reach(M, cv, ws, we, g_lo, g_hi);
    const int lane = threadIdx.x & 63;
    for (int b = 0; b < M.n_nblk; b += 64) {
        const bool in = b + lane < M.n_nblk;
        const int e = in ? M.nblk_end[b + lane] : INT32_MAX, st = in ? M.nblk_start[b + lane] : INT32_MAX;
        o0 += __popcll(__ballot(e <= g_lo));
        o1 += __popcll(__ballot(st < g_hi));
    }
}

// does the k-mer at p touch an N block?  (p outside the window's reach -- a BAM read1 poking out of its fragment is
// the one way there -- searches every block)
__device__ __attribute__((noinline)) bool motif_n_search(const int32_t* nblk_start, const int32_t* nblk_end, int n_nblk,
                                                         int k, int p, int a, int b) {
    while (a < b) {  // first block ending after p
        const int m = (a + b) >> 1;
        if (nblk_end[m] <= p) a = m + 1; else b = m;
    }
    return a < n_nblk && nblk_start[a] < p + k;
}
__device__ __forceinline__ bool motif_has_n(const MotifParams& M, int p, int o0, int o1, int g_lo, int g_hi) {
    int a = o0, b = o1;
    if (p <= g_lo || p + M.k >= g_hi) { a = 0; b = M.n_nblk; }
    if (a >= b) return false;  // nearly always
    return motif_n_search(M.nblk_start, M.nblk_end, M.n_nblk, M.k, p, a, b);
}

// The k-mer (k <= 13) at base p cut out of the four bytes that hold it, as a base-4 number in the 2bit format's OWN
// digit order (T C A G); revcomp: the reverse complement's (T<->A, C<->G is XOR 10b per digit; a bit reversal turns the
// digits around and swaps the two bits of each, put back by the pair swap).  The histogram in LDS is kept in this
// order and its bins are renamed to A C G T order (gen_kmers, utils/utils.py:388-410) when it is written out
// (motif_bin): 4^k translations per window and block instead of one per fragment end.
__device__ __forceinline__ int kmer_from_word(uint32_t w, int p, int k, bool revcomp) {
    const uint32_t be = __builtin_bswap32(w);  // first base on top
    if (!revcomp) return (int)__builtin_amdgcn_ubfe(be, 32 - 2 * ((p & 3) + k), 2 * k);
    // the field of the bit-reversed word: digits in reverse order, the two bits of each swapped
    uint32_t y = __builtin_amdgcn_ubfe(__brev(be), 2 * (p & 3), 2 * k);
    y = ((y & 0x55555555u) << 1) | ((y >> 1) & 0x55555555u);
    return (int)(y ^ (0xAAAAAAAAu & ((1u << (2 * k)) - 1u)));
}
// bin b of a T C A G ordered histogram -> its place in A C G T order (digit by digit: T=00 C=01 A=10 G=11 -> 11 01 00 10)
__device__ __forceinline__ int motif_bin(int b, int n_bins) {
    const uint32_t v1 = ((uint32_t)b >> 1) & 0x55555555u, v0 = (uint32_t)b & 0x55555555u;
    return (int)((((~(v1 ^ v0) & 0x55555555u) << 1) | (~v1 & 0x55555555u)) & (uint32_t)(n_bins - 1));
}

// What one window-feature launch computes (any combination, ONE pass over the fragments):
//   coverage count + length histogram under `wp`  (frag/_coverage.py:117-130, _frag_length.py:147-153)
//   DELFI short / long under `dp`                   (frag/_delfi.py:443-472)
struct FeatParams {
    WinPred wp;      // CH == 2 (motif pass) and the host-side description of the filter
    DelfiPred dp;    // blacklist CSR pointers (bl_*) are read from here
    MotifParams mp;  // CH == 2
    // CH == 1 / DF: sign-test constants, clamped on the host so that no difference below can overflow
    // (coordinates are < 2^30, launch_window_features)
    int ch_q, ch_min, ch_max, is_any;
    int df_q, cen0, cen1, tel0, tel1;
    int do_cov, do_hist;
    int len_lo, n_bins;
    int64_t* cov_out;
    uint32_t* hist_out;
    int64_t* over_out;
    int64_t* short_out;
    int64_t* long_out;
};

// cov / sh / lg count REJECTED elements for CH == 1 / DF (passing = processed - rejected) and passing ones
// for CH >= 2; over: motif errors (CH == 2; the length histogram keeps its overflow in an extra LDS bin).
struct FeatAcc {
    int n = 0, cov = 0, over = 0, sh = 0, lg = 0;
};

// Every predicate of the coverage / histogram / DELFI pass is a conjunction of "a >= b" terms.  Each term is
// written as a difference that is non-negative when it holds; OR-ing the differences leaves the sign bit
// clear exactly when all hold ((a & b) >= 0 is "a >= 0 or b >= 0").  That is ~2 VALU operations per term and
// no scalar mask logic or branches: the compare + s_and form of the same tests was issue-bound at
// ~100 instructions per 64 fragments.  ws is the window start clamped to >= -1, we1 = min(window stop, 2^30) - 1.
template <int CH, bool DF, bool BAM, bool BL>
__device__ __forceinline__ void feat_element(const ContigView& cv, const FeatParams& P, int idx, int fs, int fe, int q,
                                             int ws, int we1, int o0, int o1, uint32_t* h, FeatAcc& a, bool valid) {
    if (CH == 1 || DF) {
        const int len = fe - fs;
        const int mid = (int)(((unsigned)fs + (unsigned)fe) >> 1);
        const int t_mid = (mid - ws) | (we1 - mid);  // ws <= mid < we  (implies fs < we)
        const int t_lo = fe - 1 - ws;                // fe > ws
        int t_fetch;                                 // the index query that produced the stream
        if (BAM) {
            const int rs = cv.r1_start[idx], re = cv.r1_end[idx];
            t_fetch = (we1 - rs) | (re - 1 - ws);    // read1 overlaps the window (io/alignment.py:245)
        } else {
            t_fetch = t_lo;                          // with t_mid: the fragment overlaps it (:270-279)
        }
        if (CH == 1) {
            int x = (q - P.ch_q) | (len - P.ch_min) | (P.ch_max - len);
            const int t_any = t_lo | (we1 - fs);     // utils/_frag_generator.py:44-50
            x |= P.is_any ? (BAM ? (t_any | t_fetch) : t_any) : (t_mid | t_fetch);
            const unsigned bad = (unsigned)x >> 31;
            a.cov += bad;
            if (P.do_hist) {  // out-of-range lengths land in bin n_bins (the overflow count)
                const unsigned b = min((unsigned)(len - P.len_lo), (unsigned)P.n_bins);
                atomicAdd(&h[b], bad ^ 1u);
            }
        }
        if (DF) {  // frag/_delfi.py:443-472
            int y = (q - P.df_q) | (len - 100) | (220 - len) | t_mid | t_fetch;
            y |= (P.cen0 - fe) & (fs - P.cen1);      // not (fe > cen0 and fs < cen1)
            y |= (P.tel0 - fe) & (fs - P.tel1);
            if (BL) {  // blacklisted iff max{r1 : r0 <= fs, region inside the window} > fe (:455-462)
                if (y >= 0) {
                    int lo = o0, hi = o1;  // upper bound: first region with r0 > fs
                    while (lo < hi) {
                        const int m = (lo + hi) >> 1;
                        if (P.dp.bl_r0[m] <= fs) lo = m + 1; else hi = m;
                    }
                    if (lo > o0 && P.dp.bl_pm[lo - 1] > fe) y = -1;
                }
            }
            a.sh += (unsigned)(y | (150 - len)) >> 31;
            a.lg += (unsigned)(y | (len - 151)) >> 31;
        }
        a.n += 1;
    }
    if (CH >= 2) {
        // frag/_end_motifs.py:118-176, frag/_breakpoint_motifs.py:124-185: every fetched fragment
        // (index overlap + mapq only) contributes the k-mer at its start and / or the reverse
        // complement of the k-mer at its stop.  ws / we1 are the raw window bounds here.
        if (valid && P.wp.test<BAM>(cv, idx, fs, fe, q, ws, we1)) {
            ++a.cov;
            const MotifParams& M = P.mp;
            bool skip = M.guard > 0 && (fs - M.guard < 0 || fs + M.guard >= M.chrom_len);
            if (!skip) {
                const bool use_fwd = M.both || (!M.neg && cv.strand[idx] != 0);
                const bool use_rev = M.both || M.neg;
                if (use_fwd) {
                    const int lo = fs + M.f_off;
                    if (lo < 0 || lo + M.k > M.chrom_len) {
                        skip = true;  // the reference's `continue` also drops this fragment's other end
                    } else {
                        const int code = kmer_code(M, lo, false);
                        if (code >= 0) atomicAdd(&h[code], 1u);
                    }
                }
                if (!skip && use_rev) {
                    const int lo = fe + M.r_off;
                    if (lo < 0 || lo + M.k > M.chrom_len) {
                        if (M.rev_err) ++a.over;
                    } else {
                        const int code = kmer_code(M, lo, true);
                        if (code >= 0) atomicAdd(&h[code], 1u);
                    }
                }
            }
        }
    }
}

// Every condition below is a difference whose SIGN says "fails", OR-ed together like feat_element's: sixteen elements'
// worth of bool flags live as 64-bit lane masks in SGPRs, spill to VGPR lanes, and the pass turns issue-bound on the
// readlane / mask traffic (134 VALU + 79 SALU instructions per fragment in the first form of this function).
// ws / we1 are the CLAMPED bounds (window_bounds<1>); the filter is the motif pass' own (policy ANY, no length
// bounds: ftk_motif_counts), so the window test is overlap + mapq (+ read1 overlap for a BAM fetch).
// EDGE = false: the window's reach (motif_reach) lies inside the contig, so no k-mer of a FETCHED fragment can start
// outside it and the two range tests are left out (chosen per window; never for a BAM fetch, whose read1 may poke out).
template <bool BAM, int N, bool EDGE = true>
__device__ __forceinline__ void motif_prep(const ContigView& cv, const FeatParams& P, const int (&idx)[N],
                                           const int (&fs)[N], const int (&fe)[N], const int (&q)[N], int hi, int ws,
                                           int we1, int (&pf)[N], int (&pr)[N], uint32_t (&cf)[N], uint32_t (&cr)[N],
                                           uint32_t (&wf)[N], uint32_t (&wr)[N], FeatAcc& a, int& n_ok) {
    const MotifParams& M = P.mp;
    typedef uint32_t __attribute__((aligned(1))) u32u;
    const int u_rev = (M.both || M.neg) ? 0 : -1;   // the stop's k-mer is wanted at all
    const int last = M.chrom_len - M.k;              // last base a k-mer may start at
#pragma unroll
    for (int j = 0; j < N; ++j) {
        int x = (q[j] - P.ch_q) | (fe[j] - 1 - ws) | (we1 - fs[j]) | (hi - 1 - idx[j]);
        if (BAM) x |= (we1 - cv.r1_start[min(idx[j], hi - 1)]) | (cv.r1_end[min(idx[j], hi - 1)] - 1 - ws);
        n_ok += __popcll(__ballot(x >= 0));  // (scalar unit) fetched fragments of the wave
        if (M.guard > 0) x |= (fs[j] - M.guard) | (M.chrom_len - 1 - M.guard - fs[j]);
        int u_fwd = M.both ? 0 : -1;
        if (!M.both && !M.neg) u_fwd = cv.strand[min(idx[j], hi - 1)] != 0 ? 0 : -1;
        pf[j] = fs[j] + M.f_off;  // where the two k-mers start
        pr[j] = fe[j] + M.r_off;
        int no_f = x | u_fwd, no_r;
        if (EDGE) {
            const int f_out = pf[j] | (last - pf[j]), r_out = pr[j] | (last - pr[j]);
            no_f |= f_out;
            x |= ~u_fwd & f_out;  // the reference's `continue` on a start outside the contig also drops the other end
            no_r = x | u_rev | r_out;
            if (M.rev_err) a.over += (unsigned)(~(x | u_rev) & r_out) >> 31;
        } else {
            no_r = x | u_rev;
        }
        cf[j] = (uint32_t)~no_f >> 31;  // what the end adds to its bin: 1, or 0 when it contributes nothing
        cr[j] = (uint32_t)~no_r >> 31;
        wf[j] = *reinterpret_cast<const u32u*>(M.img + (min((unsigned)pf[j], (unsigned)max(last, 0)) >> 2));
        wr[j] = *reinterpret_cast<const u32u*>(M.img + (min((unsigned)pr[j], (unsigned)max(last, 0)) >> 2));
    }
}

template <bool BAM, int N>
__device__ __forceinline__ void motif_commit(const ContigView& cv, const FeatParams& P, int ws, int we1, int o0, int o1,
                                             const int (&pf)[N], const int (&pr)[N], const uint32_t (&cf)[N],
                                             const uint32_t (&cr)[N], const uint32_t (&wf)[N], const uint32_t (&wr)[N],
                                             uint32_t* h) {
    const MotifParams& M = P.mp;
    int g_lo = 0, g_hi = 0;
    const bool n_near = o1 > o0;  // (uniform) N blocks within the window's reach
    if (M.n_nblk && (BAM || n_near)) motif_reach(M, cv, ws, we1 + 1, g_lo, g_hi);
#pragma unroll
    for (int j = 0; j < N; ++j) {
        // tabix fetch: every k-mer lies within reach, so a window without N blocks nearby never searches;
        // an end that contributes nothing adds 0 to whatever bin its word names -- no branch
        uint32_t f = cf[j], r = cr[j];
        if (M.n_nblk && (BAM || n_near)) {
            if (f && motif_has_n(M, pf[j], o0, o1, g_lo, g_hi)) f = 0;
            if (r && motif_has_n(M, pr[j], o0, o1, g_lo, g_hi)) r = 0;
        }
        atomicAdd(&h[kmer_from_word(wf[j], pf[j], M.k, false)], f);
        atomicAdd(&h[kmer_from_word(wr[j], pr[j], M.k, true)], r);
    }
}

template <bool BAM, int N>
__device__ __forceinline__ void motif_elements(const ContigView& cv, const FeatParams& P, const int (&idx)[N],
                                               const int (&fs)[N], const int (&fe)[N], const int (&q)[N], int hi,
                                               int ws, int we1, int o0, int o1, uint32_t* h, FeatAcc& a) {
    int pf[N], pr[N], n_ok = 0;
    uint32_t cf[N], cr[N], wf[N], wr[N];
    motif_prep<BAM, N>(cv, P, idx, fs, fe, q, hi, ws, we1, pf, pr, cf, cr, wf, wr, a, n_ok);
    motif_commit<BAM, N>(cv, P, ws, we1, o0, o1, pf, pr, cf, cr, wf, wr, h);
    if ((threadIdx.x & 63) == 0) a.cov += n_ok;
}

// The fragments [lo, hi) of one window as a software pipeline, kBS threads: kMotifAhead slabs (4 fragments per thread
// each) of columns are in flight; a slab that has arrived is tested and its reference words requested, its column
// registers are re-filled with the slab kMotifAhead further on, and only then is the PREVIOUS slab's histogram work
// done -- so a wave always has column loads and gathers outstanding while it computes.
constexpr int kMotifAhead = 4;
template <int kBS, bool BAM, bool EDGE>
__device__ __forceinline__ void motif_stream_t(const ContigView& cv, const FeatParams& P, int lo, int hi, int tid, int ws,
                                               int we1, int o0, int o1, uint32_t* h, FeatAcc& a) {
    constexpr int D = kMotifAhead, kSlab = 4 * kBS;
    if (lo >= hi) return;  // (uniform)
    int4 s4[D], e4[D];
    uchar4 q4[D];
    const int i0 = lo + 4 * tid;
    // Loads are never conditional: a group past the end reads the range's last group again (its elements fail the
    // idx < hi test).  A load under `if (i < hi)` makes the compiler merge old and new registers right behind it --
    // with an s_waitcnt on the load it has just issued, which is the end of any prefetch.
    const int last_group = (hi - 1) & ~3;  // lo is a multiple of 4 and hi > lo
#pragma unroll
    for (int u = 0; u < D; ++u) {
        const int i = min(i0 + u * kSlab, last_group);
        s4[u] = *reinterpret_cast<const int4*>(cv.start + i);
        e4[u] = *reinterpret_cast<const int4*>(cv.end + i);
        q4[u] = *reinterpret_cast<const uchar4*>(cv.mapq + i);
    }
    int pf0[4] = {0, 0, 0, 0}, pr0[4] = {0, 0, 0, 0}, n_ok = 0;  // the slab whose words are on their way
    uint32_t cf0[4] = {0, 0, 0, 0}, cr0[4] = {0, 0, 0, 0}, wf0[4] = {0, 0, 0, 0}, wr0[4] = {0, 0, 0, 0};
    for (int slab = lo; slab < hi; slab += D * kSlab) {  // (uniform)
#pragma unroll
        for (int u = 0; u < D; ++u) {
            if (slab + u * kSlab >= hi) break;           // (uniform)
            const int i = slab + 4 * tid + u * kSlab;
            const int ii[4] = {i, i + 1, i + 2, i + 3};
            const int ss[4] = {s4[u].x, s4[u].y, s4[u].z, s4[u].w}, ee[4] = {e4[u].x, e4[u].y, e4[u].z, e4[u].w};
            const int qq[4] = {q4[u].x, q4[u].y, q4[u].z, q4[u].w};
            int pf[4], pr[4];
            uint32_t cf[4], cr[4], wf[4], wr[4];
            motif_prep<BAM, 4, EDGE>(cv, P, ii, ss, ee, qq, hi, ws, we1, pf, pr, cf, cr, wf, wr, a, n_ok);
            const int nxt = min(i + D * kSlab, last_group);
            s4[u] = *reinterpret_cast<const int4*>(cv.start + nxt);
            e4[u] = *reinterpret_cast<const int4*>(cv.end + nxt);
            q4[u] = *reinterpret_cast<const uchar4*>(cv.mapq + nxt);
            motif_commit<BAM, 4>(cv, P, ws, we1, o0, o1, pf0, pr0, cf0, cr0, wf0, wr0, h);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                pf0[j] = pf[j]; pr0[j] = pr[j]; cf0[j] = cf[j]; cr0[j] = cr[j]; wf0[j] = wf[j]; wr0[j] = wr[j];
            }
        }
    }
    motif_commit<BAM, 4>(cv, P, ws, we1, o0, o1, pf0, pr0, cf0, cr0, wf0, wr0, h);
    if ((tid & 63) == 0) a.cov += n_ok;
}

template <int kBS, bool BAM>
__device__ __forceinline__ void motif_stream(const ContigView& cv, const FeatParams& P, int lo, int hi, int tid, int ws,
                                             int we1, int o0, int o1, uint32_t* h, FeatAcc& a) {
    int g_lo, g_hi;
    motif_reach(P.mp, cv, ws, we1 + 1, g_lo, g_hi);
    if (!BAM && g_lo >= -1 && g_hi - 1 <= P.mp.chrom_len && P.mp.guard <= 0)  // (uniform) every k-mer inside the contig
        motif_stream_t<kBS, BAM, false>(cv, P, lo, hi, tid, ws, we1, o0, o1, h, a);
    else
        motif_stream_t<kBS, BAM, true>(cv, P, lo, hi, tid, ws, we1, o0, o1, h, a);
}

// Window bounds as the element tests want them.
template <int CH>
__device__ __forceinline__ void window_bounds(int ws_raw, int we_raw, int& ws, int& we1) {
    if (CH == 2) { ws = ws_raw; we1 = we_raw; return; }
    ws = max(ws_raw, -1);
    we1 = min(max(we_raw, -1), 1 << 30) - 1;
}

// Four fragments (one 16-byte load per column).  Fragments of the group outside [lo, hi) are real
// neighbours or padding (start = end = 2^30): both fail every window test, so CH == 1 / DF need no
// bounds check; the motif pass keeps one.
template <int CH, bool DF, bool BAM, bool BL>
__device__ __forceinline__ void feat_group(const ContigView& cv, const FeatParams& P, int i, int hi, const int4& s,
                                           const int4& e, const uchar4& q, int ws, int we1, int o0, int o1,
                                           uint32_t* h, FeatAcc& a) {
    const int ss[4] = {s.x, s.y, s.z, s.w}, ee[4] = {e.x, e.y, e.z, e.w}, qq[4] = {q.x, q.y, q.z, q.w};
    if (CH == kMotifWord) {  // o0 / o1: the window's N blocks
        const int ii[4] = {i, i + 1, i + 2, i + 3};
        motif_elements<BAM, 4>(cv, P, ii, ss, ee, qq, hi, ws, we1, o0, o1, h, a);
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
        feat_element<CH, DF, BAM, BL>(cv, P, i + j, ss[j], ee[j], qq[j], ws, we1, o0, o1, h, a, i + j < hi);
}

// ---------------------------------------------------------------------------
// window features, small path: one wave per window (candidate range <= kSmallMax)
// ---------------------------------------------------------------------------
template <int CH, bool DF, bool BAM>
__global__ __launch_bounds__(256) void feat_small_kernel(ContigView cv, const int32_t* ws_, const int32_t* we_,
                                                         int n_win, const int32_t* cand_lo, const int32_t* cand_hi,
                                                         const uint32_t* nchunks, FeatParams P) {
    extern __shared__ uint32_t lds_hist[];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int w = blockIdx.x * 4 + wv;
    if (w >= n_win) return;
    const bool hist = CH && P.do_hist;
    if (nchunks[w] != 0) {
        // the chunked path owns this window and adds into its histogram row with atomics: clear it here
        // (this kernel precedes feat_large_kernel on the stream), so no separate memset is needed
        if (hist) {
            uint32_t* dst = P.hist_out + (size_t)w * P.n_bins;
            for (int b = lane; b < P.n_bins; b += 64) dst[b] = 0;
            if (lane == 0) P.over_out[w] = 0;
        }
        return;
    }
    const int lo = cand_lo[w], hi = cand_hi[w];
    int ws, we1;
    window_bounds<CH>(ws_[w], we_[w], ws, we1);
    uint32_t* h = lds_hist + (size_t)wv * (P.n_bins + 1);
    if (hist && lo < hi) {
        for (int b = lane; b <= P.n_bins; b += 64) h[b] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    int o0 = 0, o1 = 0;
    if (DF && P.dp.bl_off) { o0 = P.dp.bl_off[w]; o1 = P.dp.bl_off[w + 1]; }
    if (CH == kMotifWord) motif_window(P.mp, cv, ws, we1 + 1, o0, o1);
    FeatAcc a;
    for (int i = lo + 4 * lane; i < hi; i += 256) {  // lo is a multiple of 4 (planner)
        const int4 s = *reinterpret_cast<const int4*>(cv.start + i);
        const int4 e = *reinterpret_cast<const int4*>(cv.end + i);
        const uchar4 q = *reinterpret_cast<const uchar4*>(cv.mapq + i);
        if (DF && o1 > o0) feat_group<CH, DF, BAM, true>(cv, P, i, hi, s, e, q, ws, we1, o0, o1, h, a);
        else feat_group<CH, DF, BAM, false>(cv, P, i, hi, s, e, q, ws, we1, o0, o1, h, a);
    }
    int over = 0;
    if (hist && lo < hi) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        uint32_t* dst = P.hist_out + (size_t)w * P.n_bins;
        for (int b = lane; b < P.n_bins; b += 64) dst[CH == kMotifWord ? motif_bin(b, P.n_bins) : b] = h[b];  // full row: no pre-fill needed
        if (CH == 1) over = (int)h[P.n_bins];
    } else if (hist) {
        uint32_t* dst = P.hist_out + (size_t)w * P.n_bins;
        for (int b = lane; b < P.n_bins; b += 64) dst[b] = 0;
    }
    a.n = wave_reduce_add(a.n);
    a.cov = wave_reduce_add(a.cov);
    if (CH >= 2) over = wave_reduce_add(a.over);
    a.sh = wave_reduce_add(a.sh);
    a.lg = wave_reduce_add(a.lg);
    if (lane == 0) {
        if (CH && P.do_cov) P.cov_out[w] = CH == 1 ? a.n - a.cov : a.cov;
        if (hist) P.over_out[w] = over;
        if (DF) { P.short_out[w] = a.n - a.sh; P.long_out[w] = a.n - a.lg; }
    }
}

// ---------------------------------------------------------------------------
// window features, large path: a fixed grid walks the list of 4096-fragment
// chunks in order; a block keeps accumulating while the window stays the same
// and issues one atomic per counter per (block, window).
// ---------------------------------------------------------------------------
template <int CH, bool DF, bool BAM, bool BL>
__device__ __forceinline__ void feat_chunk(const ContigView& cv, const FeatParams& P, int lo, int hi, int tid, int ws,
                                           int we1, int o0, int o1, uint32_t* h, FeatAcc& a) {
    // a whole chunk (4 x 1024 fragments) is requested before any of it is used
    const int i0 = lo + 4 * tid;
    int4 s4[4], e4[4];
    uchar4 q4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 1024;
        if (i < hi) {
            s4[u] = *reinterpret_cast<const int4*>(cv.start + i);
            e4[u] = *reinterpret_cast<const int4*>(cv.end + i);
            q4[u] = *reinterpret_cast<const uchar4*>(cv.mapq + i);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * 1024;
        if (i < hi) feat_group<CH, DF, BAM, BL>(cv, P, i, hi, s4[u], e4[u], q4[u], ws, we1, o0, o1, h, a);
    }
}

template <int CH, bool DF, bool BAM>
__global__ __launch_bounds__(256) void feat_large_kernel(ContigView cv, const int32_t* ws_, const int32_t* we_,
                                                         int n_win, const int32_t* cand_lo, const int32_t* cand_hi,
                                                         const uint32_t* chunk_off, FeatParams P) {
    extern __shared__ uint32_t lds_hist[];
    __shared__ int red[5][4];
    const uint32_t total = chunk_off[n_win];
    const uint32_t c0 = (uint32_t)(((unsigned long long)total * blockIdx.x) / gridDim.x);
    const uint32_t c1 = (uint32_t)(((unsigned long long)total * (blockIdx.x + 1)) / gridDim.x);
    if (c0 >= c1) return;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool hist = CH && P.do_hist;
    if (hist) {
        for (int b = tid; b <= P.n_bins; b += 256) lds_hist[b] = 0;
        __syncthreads();
    }
    int w;
    {
        // largest w with chunk_off[w] <= c0, 64 probes per step (each wave for itself): two or three dependent loads
        // where a bisection needs log2(n_win) -- a block that walks one or two chunks spent longer finding its window
        // than on its fragments
        int lo = 0, hi = n_win;  // chunk_off[lo] <= c0 < chunk_off[hi]
        while (hi - lo > 1) {
            const int step = (hi - lo + 63) >> 6;
            const int at = lo + (lane + 1) * step;
            const bool le = at < hi && chunk_off[at] <= c0;  // true for a prefix of the lanes (chunk_off ascends)
            lo += __popcll(__ballot(le)) * step;
            hi = min(hi, lo + step);
        }
        w = lo;
    }
    FeatAcc a;
    uint32_t c = c0;
    while (c < c1) {
        const uint32_t w_first = chunk_off[w], w_next = chunk_off[w + 1];
        if (c >= w_next) { ++w; continue; }
        int ws, we1;
        window_bounds<CH>(ws_[w], we_[w], ws, we1);
        const int wlo = cand_lo[w], whi = cand_hi[w];
        int o0 = 0, o1 = 0;
        if (DF && P.dp.bl_off) { o0 = P.dp.bl_off[w]; o1 = P.dp.bl_off[w + 1]; }
        if (CH == kMotifWord) motif_window(P.mp, cv, ws, we1 + 1, o0, o1);
        const uint32_t c_end = min(c1, w_next);
        if (CH == kMotifWord) {  // this block's chunks of the window as one stream
            const int lo = wlo + (int)(c - w_first) * kChunk;  // multiple of 4 (planner)
            const int hi = (int)min((long long)wlo + (long long)(c_end - w_first) * kChunk, (long long)whi);
            motif_stream<256, BAM>(cv, P, lo, hi, tid, ws, we1, o0, o1, lds_hist, a);
            c = c_end;
        } else
        for (; c < c_end; ++c) {
            const int lo = wlo + (int)(c - w_first) * kChunk;  // multiple of 4 (planner)
            const int hi = min(lo + kChunk, whi);
            if (DF && o1 > o0) feat_chunk<CH, DF, BAM, true>(cv, P, lo, hi, tid, ws, we1, o0, o1, lds_hist, a);
            else feat_chunk<CH, DF, BAM, false>(cv, P, lo, hi, tid, ws, we1, o0, o1, lds_hist, a);
        }
        // ---- flush this window's partial results -------------------------------------
        if (hist) {
            __syncthreads();
            uint32_t* dst = P.hist_out + (size_t)w * P.n_bins;
            for (int b = tid; b < P.n_bins; b += 256) {
                const uint32_t v = lds_hist[b];
                if (v) { atomicAdd(&dst[CH == kMotifWord ? motif_bin(b, P.n_bins) : b], v); lds_hist[b] = 0; }
            }
        }
        a.n = wave_reduce_add(a.n);
        a.cov = wave_reduce_add(a.cov);
        a.over = wave_reduce_add(a.over);
        a.sh = wave_reduce_add(a.sh);
        a.lg = wave_reduce_add(a.lg);
        if (lane == 0) { red[0][wv] = a.cov; red[1][wv] = a.over; red[2][wv] = a.sh; red[3][wv] = a.lg; red[4][wv] = a.n; }
        __syncthreads();
        if (tid < 4) {
            int t = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
            const int n = red[4][0] + red[4][1] + red[4][2] + red[4][3];
            if (tid == 1 && CH == 1 && hist) { t = (int)lds_hist[P.n_bins]; lds_hist[P.n_bins] = 0; }
            else if (tid != 1 && (CH == 1 || tid >= 2)) t = n - t;  // rejected -> passing
            int64_t* dst = tid == 0 ? (CH && P.do_cov ? P.cov_out : nullptr)
                         : tid == 1 ? (hist ? P.over_out : nullptr)
                         : tid == 2 ? (DF ? P.short_out : nullptr) : (DF ? P.long_out : nullptr);
            if (dst && t) atomicAdd(reinterpret_cast<unsigned long long*>(dst + w), (unsigned long long)t);
        }
        __syncthreads();
        a = FeatAcc{};
    }
}

// Threads per block of the block-per-window kernels: a template parameter.  512 threads keep twice the
// fragments in flight per window, which pays once a window holds several thousand candidates (43 -> 40 us per
// contig in the whole-genome bench; 128 / 64 threads: 62 / 91 us, 1024: 51 us); 256 for smaller windows.
// The whole candidate range [lo, hi) of a window as a software pipeline of slabs of 4 * kFeatBS fragments
// (one 16-byte load per column and thread): four slabs are requested up front and each slab's registers are
// re-filled with the slab four ahead as soon as it has been consumed, so three to four 16-byte
// loads per column stay in flight behind the arithmetic for the whole range.
template <int kFeatBS, int CH, bool DF, bool BAM, bool BL>
__device__ __forceinline__ void feat_stream(const ContigView& cv, const FeatParams& P, int lo, int hi, int tid, int ws,
                                            int we1, int o0, int o1, uint32_t* h, FeatAcc& a) {
    int4 s4[4], e4[4];
    uchar4 q4[4];
    const int i0 = lo + 4 * tid;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * (4 * kFeatBS);
        if (i < hi) {
            s4[u] = *reinterpret_cast<const int4*>(cv.start + i);
            e4[u] = *reinterpret_cast<const int4*>(cv.end + i);
            q4[u] = *reinterpret_cast<const uchar4*>(cv.mapq + i);
        }
    }
    for (int base = i0; base < hi; base += 16 * kFeatBS) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * (4 * kFeatBS);
            if (i < hi) {
                const int4 s = s4[u], e = e4[u];
                const uchar4 q = q4[u];
                const int nxt = i + 16 * kFeatBS;
                if (nxt < hi) {
                    s4[u] = *reinterpret_cast<const int4*>(cv.start + nxt);
                    e4[u] = *reinterpret_cast<const int4*>(cv.end + nxt);
                    q4[u] = *reinterpret_cast<const uchar4*>(cv.mapq + nxt);
                }
                feat_group<CH, DF, BAM, BL>(cv, P, i, hi, s, e, q, ws, we1, o0, o1, h, a);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// window features, block path: one block per window, no planning pass.  Used when the windows are
// many and of similar length (bin tilings): the block derives its window's candidate range from
// the 512-bp index itself, walks it in 4096-fragment chunks and OWNS the window's outputs -- plain
// stores, nothing to zero beforehand, no atomics.  One launch per call instead of three.
// ---------------------------------------------------------------------------
template <int kFeatBS, int CH, bool DF, bool BAM>
__device__ __forceinline__ void feat_block_body(const ContigView& cv, int ws_raw, int we_raw, int lmax,
                                                const FeatParams& P, int o0, int o1, size_t row,
                                                uint32_t* lds_hist, int (*red)[kFeatBS / 64]) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool hist = CH && P.do_hist;
    int lo, hi;
    uint32_t nc;
    window_candidates(cv, ws_raw, we_raw, lmax, -1, lo, hi, nc);
    if (hist) {
        for (int b = tid; b <= P.n_bins; b += kFeatBS) lds_hist[b] = 0;
        __syncthreads();
    }
    int ws, we1;
    window_bounds<CH>(ws_raw, we_raw, ws, we1);
    FeatAcc a;
    if (CH == kMotifWord) motif_window(P.mp, cv, ws, we1 + 1, o0, o1);
    if (CH == kMotifWord) motif_stream<kFeatBS, BAM>(cv, P, lo, hi, tid, ws, we1, o0, o1, lds_hist, a);
    else if (DF && o1 > o0) feat_stream<kFeatBS, CH, DF, BAM, true>(cv, P, lo, hi, tid, ws, we1, o0, o1, lds_hist, a);
    else feat_stream<kFeatBS, CH, DF, BAM, false>(cv, P, lo, hi, tid, ws, we1, o0, o1, lds_hist, a);
    if (hist) {
        __syncthreads();
        uint32_t* dst = P.hist_out + row * P.n_bins;
        for (int b = tid; b < P.n_bins; b += kFeatBS) dst[CH == kMotifWord ? motif_bin(b, P.n_bins) : b] = lds_hist[b];
    }
    a.n = wave_reduce_add(a.n);
    a.cov = wave_reduce_add(a.cov);
    a.over = wave_reduce_add(a.over);
    a.sh = wave_reduce_add(a.sh);
    a.lg = wave_reduce_add(a.lg);
    if (lane == 0) { red[0][wv] = a.cov; red[1][wv] = a.over; red[2][wv] = a.sh; red[3][wv] = a.lg; red[4][wv] = a.n; }
    __syncthreads();
    if (tid < 4) {
        int t = 0, n = 0;
#pragma unroll
        for (int k = 0; k < kFeatBS / 64; ++k) { t += red[tid][k]; n += red[4][k]; }
        if (tid == 1 && CH == 1 && hist) t = (int)lds_hist[P.n_bins];
        else if (tid != 1 && (CH == 1 || tid >= 2)) t = n - t;  // rejected -> passing
        int64_t* dst = tid == 0 ? (CH && P.do_cov ? P.cov_out : nullptr)
                     : tid == 1 ? (hist ? P.over_out : nullptr)
                     : tid == 2 ? (DF ? P.short_out : nullptr) : (DF ? P.long_out : nullptr);
        if (dst) dst[row] = t;
    }
}

template <int kFeatBS, int CH, bool DF, bool BAM>
__global__ __launch_bounds__(kFeatBS) void feat_block_kernel(ContigView cv, const int32_t* ws_, const int32_t* we_,
                                                         int n_win, int lmax, FeatParams P) {
    extern __shared__ uint32_t lds_hist[];
    __shared__ int red[5][kFeatBS / 64];
    const int w = blockIdx.x;
    int o0 = 0, o1 = 0;
    if (DF && P.dp.bl_off) { o0 = P.dp.bl_off[w]; o1 = P.dp.bl_off[w + 1]; }
    feat_block_body<kFeatBS, CH, DF, BAM>(cv, ws_[w], we_[w], lmax, P, o0, o1, (size_t)w, lds_hist, red);
}

// ---------------------------------------------------------------------------
// window features, block path, FAST form: the common request -- tabix fetch semantics, midpoint policy, no
// length bounds on the coverage / histogram filter, one mapq cut for coverage and DELFI -- with every shared
// term computed once.  The general element (feat_element) costs ~45 VALU instructions per fragment for the
// fused pass and the CU has 64 lane-operations per clock for the one fragment per clock that 5.5 TB/s feed
// it: the pass was issue-bound, not bandwidth-bound (63 us fused vs 48 us coverage-only on a chr2-sized
// contig).  Here:
//   * the midpoint test works on m2 = fs + fe against doubled bounds (floor(m2/2) >= ws <=> m2 >= 2 ws;
//     floor(m2/2) <= we-1 <=> m2 <= 2(we-1)+1): no shift;
//   * x = (q - q_min) | (m2 - 2 ws) | (2 we - 1 - m2) | (fe - 1 - ws) is the coverage predicate AND the
//     common part of the DELFI predicate: y = x | (len - 100) | (220 - len);
//   * the centromere / telomere terms are only compiled into the loop of a window that lies within reach of
//     one of those intervals (block-uniform choice, like the blacklist bisection);
//   * DELFI keeps two counters: rejected (sign of y) and passing-and-long (sign of ~y & (150 - len));
//     short = processed - rejected - long;
//   * the number of processed fragments is 4 * ceil((hi - lo) / 4) -- not counted.
// Overflow analysis (coordinates < 2^30; padding fragments sit at 2^30): bounds are clamped to ws in
// [0, 2^30] for the doubled test, we - 1 in [-1, 2^30 - 1]; m2 - 2 ws and 2 we - 1 - m2 then stay inside int32
// for every real fragment; a padding fragment (m2 = 2^31 wraps to INT32_MIN) fails the first test when
// ws = 0 and the second one otherwise (2 we - 1 - m2 wraps negative for we >= 1; we <= 0 windows have no
// candidate range at all).
// ---------------------------------------------------------------------------
struct FastAcc {
    int cov = 0, rej = 0, lg = 0;
};

struct FastWin {
    int ws2, we2, wsm1;  // 2 * max(ws, 0); 2 * (we - 1) + 1; -1 - max(ws, -1)
    int ws, we;          // BAM: the clamped bounds themselves (ws >= -1, we <= 2^30)
    int r1_all;          // BAM: -1 when the read1 columns must be read for every fragment, else 0
};

// One fragment whose window test x (sign set = rejected) is known.
template <bool CHK, bool HIST, bool DF, bool BL, bool GAPS>
__device__ __forceinline__ void fast_element(const FeatParams& P, int fs, int fe, int x, int o0, int o1, uint32_t* h,
                                             FastAcc& a) {
    const int len = fe - fs;
    if (CHK) {
        const unsigned bad = (unsigned)x >> 31;
        a.cov += bad;
        if (HIST) {  // out-of-range lengths land in bin n_bins (the overflow count)
            const unsigned b = min((unsigned)(len - P.len_lo), (unsigned)P.n_bins);
            atomicAdd(&h[b], bad ^ 1u);
        }
    }
    if (DF) {  // frag/_delfi.py:443-472
        int y = x | (len - 100) | (220 - len);
        if (GAPS) {
            y |= (P.cen0 - fe) & (fs - P.cen1);      // not (fe > cen0 and fs < cen1)
            y |= (P.tel0 - fe) & (fs - P.tel1);
        }
        if (BL) {  // blacklisted iff max{r1 : r0 <= fs, region inside the window} > fe (:455-462)
            if (y >= 0) {
                int lo = o0, hi = o1;
                while (lo < hi) {
                    const int m = (lo + hi) >> 1;
                    if (P.dp.bl_r0[m] <= fs) lo = m + 1; else hi = m;
                }
                if (lo > o0 && P.dp.bl_pm[lo - 1] > fe) y = -1;
            }
        }
        a.rej += (unsigned)y >> 31;
        a.lg += (unsigned)(~y & (150 - len)) >> 31;  // passes and len >= 151
    }
}

// Four fragments i .. i + 3 (one 16-byte load per column).  Tabix fetch: the fragment must overlap the window
// (fe > ws; fs < we follows from the midpoint).  BAM fetch (io/alignment.py:245): its READ1 must.  A fragment
// that lies inside the window holds its read1 there too when the contig's read1 spans lie inside their fragments
// (ContigView::r1_inside), so only a fragment that crosses a window bound AND passes everything else can be
// rejected by its read1: the group's two read1 loads are issued when it holds such a fragment (fragments are
// sorted by start: those are the first and last few groups of a window's range), by every group when the
// contig's spans are not known to lie inside (r1_all).
template <bool CHK, bool HIST, bool DF, bool BL, bool GAPS, bool BAM>
__device__ __forceinline__ void fast_group(const ContigView& cv, const FeatParams& P, int i, const int4& s,
                                           const int4& e, const uchar4& q, const FastWin& W, int o0, int o1,
                                           uint32_t* h, FastAcc& a) {
    const int fs[4] = {s.x, s.y, s.z, s.w}, fe[4] = {e.x, e.y, e.z, e.w}, qq[4] = {q.x, q.y, q.z, q.w};
    int x[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m2 = fs[j] + fe[j];
        x[j] = (qq[j] - P.ch_q) | (m2 - W.ws2) | (W.we2 - m2);
        if (!BAM) x[j] |= fe[j] + W.wsm1;
    }
    if (BAM) {
        int need = W.r1_all;
#pragma unroll
        for (int j = 0; j < 4; ++j) need |= ((fs[j] - W.ws) | (W.we - fe[j])) & ~x[j];
        if (need < 0) {
            const int4 rs = *reinterpret_cast<const int4*>(cv.r1_start + i);
            const int4 re = *reinterpret_cast<const int4*>(cv.r1_end + i);
            const int r0[4] = {rs.x, rs.y, rs.z, rs.w}, r1[4] = {re.x, re.y, re.z, re.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) x[j] |= (W.we - 1 - r0[j]) | (r1[j] + W.wsm1);  // rs < we and re > ws
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) fast_element<CHK, HIST, DF, BL, GAPS>(P, fs[j], fe[j], x[j], o0, o1, h, a);
}

template <int kFeatBS, bool CHK, bool HIST, bool DF, bool BL, bool GAPS, bool BAM>
__device__ __forceinline__ void fast_stream(const ContigView& cv, const FeatParams& P, int lo, int hi, int tid,
                                            const FastWin& W, int o0, int o1, uint32_t* h, FastAcc& a) {
    int4 s4[4], e4[4];
    uchar4 q4[4];
    const int i0 = lo + 4 * tid;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * (4 * kFeatBS);
        if (i < hi) {
            s4[u] = *reinterpret_cast<const int4*>(cv.start + i);
            e4[u] = *reinterpret_cast<const int4*>(cv.end + i);
            q4[u] = *reinterpret_cast<const uchar4*>(cv.mapq + i);
        }
    }
    for (int base = i0; base < hi; base += 16 * kFeatBS) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = base + u * (4 * kFeatBS);
            if (i < hi) {
                const int4 s = s4[u], e = e4[u];
                const uchar4 q = q4[u];
                const int nxt = i + 16 * kFeatBS;
                if (nxt < hi) {
                    s4[u] = *reinterpret_cast<const int4*>(cv.start + nxt);
                    e4[u] = *reinterpret_cast<const int4*>(cv.end + nxt);
                    q4[u] = *reinterpret_cast<const uchar4*>(cv.mapq + nxt);
                }
                fast_group<CHK, HIST, DF, BL, GAPS, BAM>(cv, P, i, s, e, q, W, o0, o1, h, a);
            }
        }
    }
}

template <int kFeatBS, bool CHK, bool HIST, bool DF, bool BAM>
__device__ __forceinline__ void feat_fast_body(const ContigView& cv, int ws_raw, int we_raw, int lmax,
                                               const FeatParams& P, int o0, int o1, size_t row, uint32_t* lds_hist,
                                               int (*red)[kFeatBS / 64]) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int lo, hi;
    uint32_t nc;
    window_candidates(cv, ws_raw, we_raw, lmax, -1, lo, hi, nc);
    if (HIST) {
        for (int b = tid; b <= P.n_bins; b += kFeatBS) lds_hist[b] = 0;
        __syncthreads();
    }
    FastWin W;
    {
        const int ws = min(max(ws_raw, -1), 1 << 30);
        const int we1 = min(max(we_raw, 0), 1 << 30) - 1;
        W.ws2 = (int)(2u * (unsigned)max(ws, 0));
        W.we2 = 2 * we1 + 1;
        W.wsm1 = -1 - ws;
        W.ws = ws;
        W.we = we1 + 1;
        W.r1_all = cv.r1_inside ? 0 : -1;
    }
    // a DELFI fragment of this window (midpoint inside, 100 <= len <= 220) lies within 110 bp of it: the gap
    // terms can only matter when the window, widened by a safe 256 bp, touches one of the two intervals
    bool gaps = false;
    if (DF) {
        const long long a0 = (long long)ws_raw - 256, a1 = (long long)we_raw + 256;
        gaps = (a1 > P.cen0 && a0 < P.cen1) || (a1 > P.tel0 && a0 < P.tel1);
    }
    FastAcc a;
    const bool bl = DF && o1 > o0;
    if (bl) {
        if (gaps) fast_stream<kFeatBS, CHK, HIST, DF, true, true, BAM>(cv, P, lo, hi, tid, W, o0, o1, lds_hist, a);
        else fast_stream<kFeatBS, CHK, HIST, DF, true, false, BAM>(cv, P, lo, hi, tid, W, o0, o1, lds_hist, a);
    } else {
        if (gaps) fast_stream<kFeatBS, CHK, HIST, DF, false, true, BAM>(cv, P, lo, hi, tid, W, o0, o1, lds_hist, a);
        else fast_stream<kFeatBS, CHK, HIST, DF, false, false, BAM>(cv, P, lo, hi, tid, W, o0, o1, lds_hist, a);
    }
    if (HIST) {
        __syncthreads();
        uint32_t* dst = P.hist_out + row * P.n_bins;
        for (int b = tid; b < P.n_bins; b += kFeatBS) dst[b] = lds_hist[b];
    }
    a.cov = wave_reduce_add(a.cov);
    a.rej = wave_reduce_add(a.rej);
    a.lg = wave_reduce_add(a.lg);
    if (lane == 0) { red[0][wv] = a.cov; red[1][wv] = a.rej; red[2][wv] = a.lg; }
    __syncthreads();
    if (tid < 4) {
        const int n = (hi - lo + 3) & ~3;  // fragments processed by the block: whole groups of four
        int cov = 0, rej = 0, lg = 0;
#pragma unroll
        for (int k = 0; k < kFeatBS / 64; ++k) { cov += red[0][k]; rej += red[1][k]; lg += red[2][k]; }
        if (tid == 0 && CHK && P.do_cov) P.cov_out[row] = n - cov;
        if (tid == 1 && HIST) P.over_out[row] = (int)lds_hist[P.n_bins];
        if (tid == 2 && DF) P.short_out[row] = n - rej - lg;
        if (tid == 3 && DF) P.long_out[row] = lg;
    }
}

template <int kFeatBS, bool CHK, bool HIST, bool DF, bool BAM>
__global__ __launch_bounds__(kFeatBS) void feat_fast_kernel(ContigView cv, const int32_t* ws_, const int32_t* we_,
                                                        int n_win, int lmax, FeatParams P) {
    extern __shared__ uint32_t lds_hist[];
    __shared__ int red[5][kFeatBS / 64];
    const int w = blockIdx.x;
    int o0 = 0, o1 = 0;
    if (DF && P.dp.bl_off) { o0 = P.dp.bl_off[w]; o1 = P.dp.bl_off[w + 1]; }
    feat_fast_body<kFeatBS, CHK, HIST, DF, BAM>(cv, ws_[w], we_[w], lmax, P, o0, o1, (size_t)w, lds_hist, red);
}

template <int kFeatBS, bool CHK, bool HIST, bool DF, bool BAM>
__global__ __launch_bounds__(kFeatBS) void feat_fast_batch_kernel(const FeatItem* __restrict__ items, int n_items,
                                                              FeatParams P) {
    extern __shared__ uint32_t lds_hist[];
    __shared__ int red[5][kFeatBS / 64];
    const int gw = blockIdx.x;
    int it = 0;
    {
        int lo = 0, hi = n_items;  // largest item with win_base <= gw
        while (hi - lo > 1) {
            const int m = (lo + hi) >> 1;
            if (items[m].win_base <= gw) lo = m; else hi = m;
        }
        it = lo;
    }
    const FeatItem& I = items[it];
    const int w = gw - I.win_base;
    FeatParams Q = P;
    Q.cen0 = I.cen0; Q.cen1 = I.cen1; Q.tel0 = I.tel0; Q.tel1 = I.tel1;
    Q.dp.bl_r0 = I.bl_r0;
    Q.dp.bl_pm = I.bl_pm;
    int o0 = 0, o1 = 0;
    if (DF && I.bl_off) { o0 = I.bl_off[w]; o1 = I.bl_off[w + 1]; }
    const ContigView cv = I.cv;
    feat_fast_body<kFeatBS, CHK, HIST, DF, BAM>(cv, I.ws[w], I.we[w], I.lmax, Q, o0, o1, (size_t)gw, lds_hist, red);
}

// The same for the windows of SEVERAL contigs in one launch (ftk_window_features_batch): block b owns
// window b of the concatenated list; its item (contig view, windows, blacklist CSR, gap constants) is
// found by bisection on the items' first rows; outputs are indexed by the global row.
template <int kFeatBS, int CH, bool DF, bool BAM>
__global__ __launch_bounds__(kFeatBS) void feat_batch_kernel(const FeatItem* __restrict__ items, int n_items,
                                                         FeatParams P) {
    extern __shared__ uint32_t lds_hist[];
    __shared__ int red[5][kFeatBS / 64];
    const int gw = blockIdx.x;
    int it = 0;
    {
        int lo = 0, hi = n_items;  // largest item with win_base <= gw
        while (hi - lo > 1) {
            const int m = (lo + hi) >> 1;
            if (items[m].win_base <= gw) lo = m; else hi = m;
        }
        it = lo;
    }
    const FeatItem& I = items[it];
    const int w = gw - I.win_base;
    FeatParams Q = P;
    Q.cen0 = I.cen0; Q.cen1 = I.cen1; Q.tel0 = I.tel0; Q.tel1 = I.tel1;
    Q.dp.bl_r0 = I.bl_r0;
    Q.dp.bl_pm = I.bl_pm;
    int o0 = 0, o1 = 0;
    if (DF && I.bl_off) { o0 = I.bl_off[w]; o1 = I.bl_off[w + 1]; }
    const ContigView cv = I.cv;
    feat_block_body<kFeatBS, CH, DF, BAM>(cv, I.ws[w], I.we[w], I.lmax, Q, o0, o1, (size_t)gw, lds_hist, red);
}

// ---------------------------------------------------------------------------
// WPS (frag/_wps.py:25-53,156-188): LDS difference array + scan per tile
// ---------------------------------------------------------------------------
// Window of "virtual" position v is [v - hl, v + hr].  Even W: hl = W/2,
// hr = W/2 - 1 and every base is its own virtual position.  Odd W = 2k+1:
// numpy.rint rounds c -/+ W/2 half-to-even, which gives base c the symmetric
// window (hl = hr = k) of v = c when (c - k) is even and of v = c - 1 when it
// is odd.  A fragment [fs, fe) contributes -1 on [fs-hr, fs+hl] and
// [fe-hr, fe+hl] (start / stop inside the window; one interval when they
// touch or overlap) and +1 on [fs+hl+1, fe-hr-1] (spanning).
// Kernel shape: blocks walk a few consecutive 4096-base tiles; the next tile's
// fragments are prefetched into registers while the current tile's scores
// stream out; the wave scan runs on DPP (no LDS traffic); every lane owns two
// adjacent bases of each 128-base half so each store instruction writes 1 KB
// contiguously (measured +18 % over 32-byte-strided stores).
// Bijective remap of a block id so that the blocks an XCD receives (id % 8 == const) cover one
// contiguous range of work items (cdna_hip_programming.md T1; placement only affects speed).
__device__ __forceinline__ long long xcd_contiguous(unsigned orig, unsigned nwg) {
    const unsigned q = nwg / 8, r = nwg % 8, xcd = orig % 8;
    const unsigned base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return (long long)base + orig / 8;
}

// inclusive scan across the 64 lanes of a wave (row_shr 1/2/4/8 inside rows of
// 16, then row_bcast:15 / row_bcast:31 across rows)
__device__ __forceinline__ int wave_incl_scan_dpp(int x) {
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}

struct WpsTile {
    long long t0;        // first base
    long long fmin, fmax;  // fetch window of the owning interval (frag/_wps.py:156-157)
    long long out_base;  // index of t0's score in the output
    int len_t;           // bases in the tile (<= kWpsTile)
};

constexpr int kWpsPrefetch = 4;  // fragments per thread held in registers for the next tile

// MULTI: a block walks several tiles and prefetches the next one's fragments.  BATCH: tiles of several
// contigs in one launch (one tile per block; the kernel-argument view / limits are replaced by the item's).
// FUSED: the tile's block also runs the window features (coverage, length histogram, DELFI) of every
// fragment that STARTS in its tile against a regular bin tiling, so the whole-contig pass reads the
// fragment columns once (BASELINE config 5: all features in a single pass).  A fragment's bin is the one
// holding its midpoint: the bin of the tile's first base or the next one (bins are longer than a tile plus
// the longest fragment), counted in LDS / registers for the first and with global atomics for the second.
// The block's work as a device function (block `block_id` of `n_blocks`), so that a launch may put other blocks
// in front of the WPS tiles (feat_then_wps_kernel below).
// NT: the scores leave with non-temporal 16-byte stores.  A compile-time switch on purpose: as a run-time
// `if (p.nt_store) nt-store else store` the two branches differ only in the !nontemporal metadata, and the optimiser
// may sink them into ONE plain store (it did once this body became a device function: WPS 175 -> 197 us per launch,
// the feature pass behind it 37 -> 53 us).
template <bool MULTI, bool BATCH, bool FUSED, bool NT>
__device__ __forceinline__ void wps_block(const unsigned block_id, const unsigned n_blocks, ContigView cv, WpsParams p,
                                          const int64_t* iv_start_, const int64_t* iv_stop_, const int64_t* out_off_,
                                          const int32_t* tile_iv, const int32_t* tile_k, long long n_tiles,
                                          int tiles_per_block, int64_t* __restrict__ out,
                                          const WpsItem* __restrict__ items, int n_items, const FusedParams& F) {
    constexpr int T = kWpsTile, NP = T / 1024, PF = kWpsPrefetch;
    extern __shared__ uint32_t fhist[];  // FUSED with a histogram: n_bins + 1 counters of the tile's first bin
    __shared__ int fcnt[4];              // FUSED: rejected (coverage, short, long) and processed fragments
    __shared__ __attribute__((aligned(16))) int d[T];
    __shared__ int pre_s[2];
    __shared__ int rng_s[2];
    __shared__ int wtot[NP][4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int hl = p.hl, hr = p.hr;

    // Optional XCD-contiguous block -> tile map (blocks are dealt round-robin to the 8 XCDs; this gives
    // each XCD one contiguous run of tiles so halos share an L2).  Off by default: measured slower.
    const long long bid = p.xcd_remap ? xcd_contiguous(block_id, n_blocks) : (long long)block_id;
    const long long tfirst = bid * tiles_per_block;
    const long long tlast = min(tfirst + (long long)tiles_per_block, n_tiles);
    if (tfirst >= tlast) return;

    // batched launch (ftk_wps_batch, one tile per block): the tile's interval, its contig's columns and
    // limits come from the item that owns it (bisection on the items' first tiles)
    long long b_start = 0, b_stop = 0, b_off = 0, b_tile0 = 0;
    if (BATCH) {
        int lo = 0, hi = n_items;
        while (hi - lo > 1) {
            const int m = (lo + hi) >> 1;
            if (items[m].tile_base <= tfirst) lo = m; else hi = m;
        }
        const WpsItem& I = items[lo];
        cv = I.cv;
        p.chrom_size = I.chrom_size;
        p.lmax = I.lmax;
        b_start = I.start; b_stop = I.stop; b_off = I.out_off; b_tile0 = I.tile_base;
    }

    auto tile_info = [&](long long t) {
        WpsTile ti;
        long long iv_start, iv_stop, out_off, k;
        if (BATCH) {
            iv_start = b_start; iv_stop = b_stop; out_off = b_off; k = t - b_tile0;
        } else if (tile_iv) {
            const int iv = tile_iv[t];
            iv_start = iv_start_[iv]; iv_stop = iv_stop_[iv]; out_off = out_off_[iv]; k = tile_k[t];
        } else {
            iv_start = p.start; iv_stop = p.stop; out_off = 0; k = t;
        }
        ti.t0 = iv_start + k * T;
        ti.len_t = (int)(min(ti.t0 + (long long)T, iv_stop) - ti.t0);
        ti.fmin = max(iv_start - (long long)p.max_len, 0LL);
        ti.fmax = min(iv_stop + (long long)p.max_len, p.chrom_size);
        ti.out_base = out_off + k * T;
        return ti;
    };
    // Candidate fragments: fs - hr <= t1 - 1 and fe + hl >= t0 - 1, i.e.
    // t0 - 1 - hl - lmax <= fs < t1 + hr.  Bounds come straight from the 512-bp
    // index (conservative): surplus candidates add nothing, their events clip.
    auto cand_bound = [&](const WpsTile& ti, int which) -> int {
        const long long q = which == 0 ? ti.t0 - 1 - hl - (long long)p.lmax : ti.t0 + ti.len_t + hr;
        if (q <= 0) return 0;
        const long long kb = q >> kBinShift;
        return kb >= cv.n_bins ? cv.n : cv.bin_idx[kb + which];
    };
    // one fragment -> its +-1/+-2 events in the tile's difference array
    auto apply = [&](const WpsTile& ti, int par, int i, int fs, int fe, int q) {
        const int len = fe - fs;
        const long long mid = ((long long)fs + (long long)fe) >> 1;
        if (q < p.mapq_min || len < p.min_len || len > p.max_len || mid < ti.fmin || mid >= ti.fmax) return;
        if (cv.r1_start) {
            // BAM: the fetch returns read1 alignments overlapping [fmin, fmax).  A fragment inside the fetch window
            // holds its read1 there (ContigView::r1_inside): the read1 columns are read for the fragments that
            // cross its bounds - none at all in a whole-contig call.
            if (!cv.r1_inside || (long long)fs < ti.fmin || (long long)fe > ti.fmax) {
                if (!((long long)cv.r1_start[i] < ti.fmax && (long long)cv.r1_end[i] > ti.fmin)) return;
            }
        } else if (!((long long)fs < ti.fmax && (long long)fe > ti.fmin)) {
            return;
        }
        const long long a = (long long)fs - hr - ti.t0;
        const long long b = (long long)fs + hl + 1 - ti.t0;
        const long long c = (long long)fe - hr - ti.t0;
        const long long e = (long long)fe + hl + 1 - ti.t0;
        if (e <= -1 || a >= ti.len_t) return;  // no effect on [t0 - 1, t1)
        if (a < 0) atomicAdd(&pre_s[par], -1); else if (a < T) atomicAdd(&d[a], -1);
        if (c >= b) {  // start / stop ranges disjoint: spanning range in between
            if (b < 0) atomicAdd(&pre_s[par], 2); else if (b < T) atomicAdd(&d[b], 2);
            if (c < 0) atomicAdd(&pre_s[par], -2); else if (c < T) atomicAdd(&d[c], -2);
        }
        if (e < 0) atomicAdd(&pre_s[par], 1); else if (e < T) atomicAdd(&d[e], 1);
    };

    // ---- FUSED: window features of the fragments that start in this tile -------------------------
    int f_w0 = 0, f_b1 = 0;              // first bin of the tile, start of the next bin
    int f_n = 0, f_cov = 0, f_sh = 0, f_lg = 0;  // processed / rejected counters of bin f_w0 (registers)
    auto feature = [&](long long t0, int len_t, int i, int fs, int fe, int q) {
        if ((long long)fs < t0 || (long long)fs >= t0 + len_t) return;  // another tile owns this fragment
        const int len = fe - fs;
        const int mid = (int)(((unsigned)fs + (unsigned)fe) >> 1);
        const int w = f_w0 + (mid >= f_b1);
        if (w < 0 || w >= F.n_win) return;
        const int ws = F.win_start + w * F.win_len;  // mid is inside [ws, ws + win_len) by construction
        int t_lo = fe - 1 - ws;                       // fe > ws (fs < we follows from the midpoint)
        if (F.bam) {  // read1 fetch (io/alignment.py:245): read1 overlaps the bin; only a fragment that crosses a bound can fail
            t_lo = 0;
            const int we = ws + F.win_len;
            if (!cv.r1_inside || fs < ws || fe > we) t_lo = (we - 1 - cv.r1_start[i]) | (cv.r1_end[i] - 1 - ws);
        }
        const bool primary = w == f_w0;
        if (F.do_cov | F.do_hist) {
            const int x = (q - F.ch_q) | (len - F.ch_min) | (F.ch_max - len) | t_lo;
            const unsigned bad = (unsigned)x >> 31;
            if (primary) {
                f_cov += bad;
                if (F.do_hist) atomicAdd(&fhist[min((unsigned)(len - F.len_lo), (unsigned)F.n_bins)], bad ^ 1u);
            } else if (!bad) {
                if (F.do_cov) atomicAdd(reinterpret_cast<unsigned long long*>(F.cov_out + w), 1ull);
                if (F.do_hist) {
                    const unsigned b = (unsigned)(len - F.len_lo);
                    if (b < (unsigned)F.n_bins) atomicAdd(F.hist_out + (size_t)w * F.n_bins + b, 1u);
                    else atomicAdd(reinterpret_cast<unsigned long long*>(F.over_out + w), 1ull);
                }
            }
        }
        if (F.do_delfi) {
            int y = (q - F.df_q) | (len - 100) | (220 - len) | t_lo;
            y |= (F.cen0 - fe) & (fs - F.cen1);
            y |= (F.tel0 - fe) & (fs - F.tel1);
            if (F.bl_off && y >= 0) {
                const int o0 = F.bl_off[w], o1 = F.bl_off[w + 1];
                if (o1 > o0) {
                    int lo2 = o0, hi2 = o1;
                    while (lo2 < hi2) {
                        const int m = (lo2 + hi2) >> 1;
                        if (F.bl_r0[m] <= fs) lo2 = m + 1; else hi2 = m;
                    }
                    if (lo2 > o0 && F.bl_pm[lo2 - 1] > fe) y = -1;
                }
            }
            const unsigned sb = (unsigned)(y | (150 - len)) >> 31, lb = (unsigned)(y | (len - 151)) >> 31;
            if (primary) { f_sh += sb; f_lg += lb; }
            else {
                if (!sb) atomicAdd(reinterpret_cast<unsigned long long*>(F.short_out + w), 1ull);
                if (!lb) atomicAdd(reinterpret_cast<unsigned long long*>(F.long_out + w), 1ull);
            }
        }
        if (primary) f_n += 1;
    };

    WpsTile cur = tile_info(tfirst);
    if (FUSED) {
        const long long rel = cur.t0 - (long long)F.win_start;
        f_w0 = rel >= 0 ? (int)(rel / F.win_len) : -1;
        f_b1 = F.win_start + (f_w0 + 1) * F.win_len;
        if (F.do_hist)
            for (int b = tid; b <= F.n_bins; b += 256) fhist[b] = 0;
        if (tid < 4) fcnt[tid] = 0;
    }
    // The first tile's candidate bounds come from two threads (vector loads) through LDS, behind the barrier the clearing
    // needs anyway.  Tried in round 5 (tools/experiments/wps_bounds_scalar.patch): every thread reading the two index
    // entries itself at block-uniform addresses - scalar loads, no hand-over - 4.57-4.60 -> 4.73-4.81 ms per
    // whole-genome step (0.846 -> 0.813 of peak): four waves' scalar loads per tile through the scalar cache, and their
    // s_waitcnt also waits for the LDS writes of the clearing.
    if (tid < 2) rng_s[tid] = cand_bound(cur, tid);
    if (tid == 2) { pre_s[0] = 0; pre_s[1] = 0; }
    {
        const int4 z = make_int4(0, 0, 0, 0);
        int4* d4 = reinterpret_cast<int4*>(d);
#pragma unroll
        for (int j = 0; j < T / 4 / 256; ++j) d4[j * 256 + tid] = z;
    }
    __syncthreads();
    int lo = rng_s[0], hi = rng_s[1];
    int pfs[PF], pfe[PF], pfq[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const int i = lo + tid + 256 * k;
        const bool ok = i < hi;
        pfs[k] = ok ? cv.start[i] : 0;
        pfe[k] = ok ? cv.end[i] : 0;
        pfq[k] = ok ? (int)cv.mapq[i] : -1;
    }

    const int odd = p.odd, kk = p.hl;  // odd W: hl == k
    for (long long t = tfirst; t < tlast; ++t) {
        const int par = (int)((t - tfirst) & 1);
        const bool has_next = MULTI && (t + 1 < tlast);
        WpsTile nxt = cur;
        int nb = 0;
        if (has_next) {
            nxt = tile_info(t + 1);
            if (tid < 2) nb = cand_bound(nxt, tid);
        }
        // ---- events -----------------------------------------------------------------
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int i = lo + tid + 256 * k;
            if (i < hi) {
                apply(cur, par, i, pfs[k], pfe[k], pfq[k]);
                if (FUSED) feature(cur.t0, cur.len_t, i, pfs[k], pfe[k], pfq[k]);
            }
        }
        for (int i = lo + PF * 256 + tid; i < hi; i += 256) {
            const int fs = cv.start[i], fe = cv.end[i], q = cv.mapq[i];
            apply(cur, par, i, fs, fe, q);
            if (FUSED) feature(cur.t0, cur.len_t, i, fs, fe, q);
        }
        if (FUSED) {  // counters of the tile's first bin: one LDS atomic per wave and counter
            const int a0 = wave_reduce_add(f_cov), a1 = wave_reduce_add(f_sh), a2 = wave_reduce_add(f_lg),
                      a3 = wave_reduce_add(f_n);
            if (lane == 0) { atomicAdd(&fcnt[0], a0); atomicAdd(&fcnt[1], a1); atomicAdd(&fcnt[2], a2); atomicAdd(&fcnt[3], a3); }
        }
        __syncthreads();
        if (FUSED && f_w0 >= 0 && f_w0 < F.n_win) {  // hand the first bin's partial results over (fire and forget)
            typedef unsigned long long ull;
            if (F.do_hist) {
                uint32_t* dst = F.hist_out + (size_t)f_w0 * F.n_bins;
                for (int b = tid; b < F.n_bins; b += 256) {
                    const uint32_t v = fhist[b];
                    if (v) atomicAdd(&dst[b], v);
                }
                if (tid == 0 && fhist[F.n_bins]) atomicAdd(reinterpret_cast<ull*>(F.over_out + f_w0), (ull)fhist[F.n_bins]);
            }
            if (tid == 1 && F.do_cov && fcnt[3] - fcnt[0]) atomicAdd(reinterpret_cast<ull*>(F.cov_out + f_w0), (ull)(fcnt[3] - fcnt[0]));
            if (tid == 2 && F.do_delfi && fcnt[3] - fcnt[1]) atomicAdd(reinterpret_cast<ull*>(F.short_out + f_w0), (ull)(fcnt[3] - fcnt[1]));
            if (tid == 3 && F.do_delfi && fcnt[3] - fcnt[2]) atomicAdd(reinterpret_cast<ull*>(F.long_out + f_w0), (ull)(fcnt[3] - fcnt[2]));
        }
        // ---- read the difference array (and clear it for the next tile), scan ---------
        int2 va[NP], vb[NP];
        int exa[NP], exb[NP];
        {
            int2* d2 = reinterpret_cast<int2*>(d);
            const int2 z = make_int2(0, 0);
#pragma unroll
            for (int j = 0; j < NP; ++j) {
                const int ia = j * 512 + wv * 128 + lane;  // int2 index of bases sb + 2l, sb + 2l + 1
                va[j] = d2[ia];
                vb[j] = d2[ia + 64];
                d2[ia] = z;
                d2[ia + 64] = z;
                const int sa = va[j].x + va[j].y, sb2 = vb[j].x + vb[j].y;
                const int ia_incl = wave_incl_scan_dpp(sa);
                const int ib_incl = wave_incl_scan_dpp(sb2);
                const int tot_a = __builtin_amdgcn_readlane(ia_incl, 63);
                const int tot_b = __builtin_amdgcn_readlane(ib_incl, 63);
                exa[j] = ia_incl - sa;
                exb[j] = tot_a + ib_incl - sb2;
                if (lane == 0) wtot[j][wv] = tot_a + tot_b;
            }
        }
        if (tid < 2) rng_s[tid] = nb;
        if (tid == 2) pre_s[par ^ 1] = 0;
        __syncthreads();
        // ---- prefetch the next tile's fragments (in flight behind the stores below) ---
        int base = pre_s[par];
        int lo2 = lo, hi2 = lo;
        int nfs[PF], nfe[PF], nfq[PF];
        if (has_next) { lo2 = rng_s[0]; hi2 = rng_s[1]; }
#pragma unroll
        for (int k = 0; k < PF; ++k) {
            const int i = lo2 + tid + 256 * k;
            const bool ok = MULTI && i < hi2;
            nfs[k] = ok ? cv.start[i] : 0;
            nfe[k] = ok ? cv.end[i] : 0;
            nfq[k] = ok ? (int)cv.mapq[i] : -1;
        }
        // ---- scores -------------------------------------------------------------------
        int64_t* dst = out + cur.out_base;
        const bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            int carry = base;
#pragma unroll
            for (int w2 = 0; w2 < 4; ++w2) {
                const int tt = wtot[j][w2];
                if (w2 < wv) carry += tt;
                base += tt;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int2 v = h ? vb[j] : va[j];
                const int ex = carry + (h ? exb[j] : exa[j]);  // G(base before this lane's pair)
                const int g0 = ex + v.x, g1 = g0 + v.y;
                const int i0 = j * 1024 + wv * 256 + h * 128 + 2 * lane;
                long long o0 = g0, o1 = g1;
                if (odd) {  // base c uses virtual position c - ((c - k) & 1)
                    if ((cur.t0 + i0 - kk) & 1) o0 = ex; else o1 = g0;
                }
                if (i0 + 1 < cur.len_t) {
                    if (vec_ok) {
                        typedef long long ll2 __attribute__((ext_vector_type(2)));
                        ll2 v2 = {o0, o1};
                        if (NT) __builtin_nontemporal_store(v2, reinterpret_cast<ll2*>(dst + i0));
                        else *reinterpret_cast<ll2*>(dst + i0) = v2;
                    } else {
                        dst[i0] = o0;
                        dst[i0 + 1] = o1;
                    }
                } else if (i0 < cur.len_t) {
                    dst[i0] = o0;
                }
            }
        }
        cur = nxt;
        lo = lo2;
        hi = hi2;
#pragma unroll
        for (int k = 0; k < PF; ++k) { pfs[k] = nfs[k]; pfe[k] = nfe[k]; pfq[k] = nfq[k]; }
    }
}

template <bool MULTI, bool BATCH, bool FUSED, bool NT>
__global__ __launch_bounds__(256) void wps_stream_kernel(ContigView cv, WpsParams p, const int64_t* iv_start_,
                                                         const int64_t* iv_stop_, const int64_t* out_off_,
                                                         const int32_t* tile_iv, const int32_t* tile_k,
                                                         long long n_tiles, int tiles_per_block,
                                                         int64_t* __restrict__ out,
                                                         const WpsItem* __restrict__ items, int n_items,
                                                         FusedParams F) {
    wps_block<MULTI, BATCH, FUSED, NT>(blockIdx.x, gridDim.x, cv, p, iv_start_, iv_stop_, out_off_, tile_iv, tile_k, n_tiles,
                                   tiles_per_block, out, items, n_items, F);
}

// ONE launch for a contig's step: the grid holds the window-feature blocks FIRST (block w = window w: the FAST
// block path, feat_fast_body with 256 threads) and the WPS tiles behind them (block n_win + t = tile t).  Blocks
// are dispatched in index order, so the feature pass starts alone, WPS tiles fill the chip as feature blocks
// retire (the feature pass's tail and the WPS ramp overlap instead of adding up), and WPS finds the contig's
// columns in the Infinity Cache behind the feature blocks that just read them.  The two kinds of block share
// nothing: results are those of the two separate launches.
template <bool CHK, bool HIST, bool DF, bool BAM, bool NT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(5, 5))) void feat_then_wps_kernel(ContigView cv, const int32_t* ws_, const int32_t* we_,
                                                            int n_win, int lmax, FeatParams P, WpsParams p,
                                                            long long n_tiles, int64_t* __restrict__ out) {
    if (blockIdx.x < (unsigned)n_win) {
        extern __shared__ uint32_t lds_hist[];
        __shared__ int red[5][4];
        const int w = blockIdx.x;
        int o0 = 0, o1 = 0;
        if (DF && P.dp.bl_off) { o0 = P.dp.bl_off[w]; o1 = P.dp.bl_off[w + 1]; }
        feat_fast_body<256, CHK, HIST, DF, BAM>(cv, ws_[w], we_[w], lmax, P, o0, o1, (size_t)w, lds_hist, red);
        return;
    }
    wps_block<false, false, false, NT>(blockIdx.x - (unsigned)n_win, gridDim.x - (unsigned)n_win, cv, p, nullptr, nullptr,
                                   nullptr, nullptr, nullptr, n_tiles, 1, out, nullptr, 0, FusedParams{});
}

// ---------------------------------------------------------------------------
// Cleavage profile (frag/_cleavage_profile.py:33-90,204-216): per base, the
// number of fragment ends (start of + fragments, stop of - fragments) over the
// fragment depth, as a percentage.  Same tile skeleton as WPS: LDS difference
// array for the depth + LDS counters for the ends, DPP scan, 1 KB-contiguous
// float64 stores.  Fragments are selected like frag_array(start, stop, "any").
// ---------------------------------------------------------------------------
// One (sub-)tile of TT bases starting at t0.  NARROW: the two LDS arrays hold 16-bit counters, two to a 32-bit word
// (element 2i in the low half, 2i + 1 in the high half; an atomic add of +-1 or +-65536; the low half read back as
// int16, the high half as (word - low) >> 16, which undoes the borrow a negative low half takes) - exact while every
// counter stays inside int16, which fewer than 32 768 candidate fragments guarantee.  8 KB per array instead of 16:
// the tile of 4 096 bases fits in 16 KB and a CU holds eight blocks instead of four.
template <bool NARROW, int TT>
__device__ __forceinline__ void cleave_tile(const ContigView& cv, const CleaveParams& p, long long iv_start, long long iv_stop,
                                            long long t0, int len_t, int lo, int hi, double* __restrict__ dst, int* dd,
                                            int* en, int& pre_s, int (*wtot)[4]) {
    // (the caller has cleared dd / en / pre_s and passed a barrier)
    constexpr int NP = TT / 1024;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    auto add = [&](int* arr, int idx, int v) {
        if (NARROW) atomicAdd(&arr[idx >> 1], (idx & 1) ? v * 65536 : v);
        else atomicAdd(&arr[idx], v);
    };
    auto apply = [&](int i, int fs, int fe, int q, int fwd) {
        const int len = fe - fs;
        if (q < p.mapq_min || len < p.min_len || len > p.max_len) return;
        if (!((long long)fe > iv_start && (long long)fs < iv_stop)) return;  // "any" policy (= tabix overlap)
        if (cv.r1_start && (!cv.r1_inside || (long long)fs < iv_start || (long long)fe > iv_stop) &&
            !((long long)cv.r1_start[i] < iv_stop && (long long)cv.r1_end[i] > iv_start)) return;
        const long long a = (long long)fs - t0, b = (long long)fe - t0;
        if (b > 0 && a < len_t) {  // covers [max(a,0), min(b,len_t))
            if (a <= 0) atomicAdd(&pre_s, 1); else add(dd, (int)a, 1);
            if (b < len_t) add(dd, (int)b, -1);
        }
        const long long e = fwd ? a : b;
        if (e >= 0 && e < len_t) add(en, (int)e, 1);
    };
    // the tile's candidates, four columns each, requested in ONE batch (round 5: until then a thread loaded a fragment's
    // start / end / mapq, filtered, and only then asked for its strand - two dependent trips to HBM per fragment and
    // loop turn, and the block's life is a chain of such trips: 0.53 -> see DESIGN 3.4)
    constexpr int PF = 4;
    int ps[PF], pe[PF], pq[PF], pw[PF];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const int i = lo + tid + 256 * k;
        const bool ok = i < hi;
        ps[k] = ok ? cv.start[i] : 0;
        pe[k] = ok ? cv.end[i] : 0;
        pq[k] = ok ? (int)cv.mapq[i] : -1;
        pw[k] = ok ? (int)cv.strand[i] : 0;
    }
#pragma unroll
    for (int k = 0; k < PF; ++k) {
        const int i = lo + tid + 256 * k;
        if (i < hi) apply(i, ps[k], pe[k], pq[k], pw[k]);
    }
    for (int i = lo + PF * 256 + tid; i < hi; i += 256) apply(i, cv.start[i], cv.end[i], cv.mapq[i], cv.strand[i]);
    __syncthreads();
    // element pair (2i, 2i + 1) of an array, whichever way it is stored
    auto pair_at = [&](const int* arr, int i) -> int2 {
        if (NARROW) {
            const int w = arr[i];
            const int x = (int)(short)(w & 0xffff);
            return make_int2(x, (w - x) >> 16);
        }
        return reinterpret_cast<const int2*>(arr)[i];
    };
    int2 va[NP], vb[NP];
    int exa[NP], exb[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int ia = j * 512 + wv * 128 + lane;
        va[j] = pair_at(dd, ia);
        vb[j] = pair_at(dd, ia + 64);
        const int sa = va[j].x + va[j].y, sb2 = vb[j].x + vb[j].y;
        const int ia_incl = wave_incl_scan_dpp(sa);
        const int ib_incl = wave_incl_scan_dpp(sb2);
        const int tot_a = __builtin_amdgcn_readlane(ia_incl, 63);
        const int tot_b = __builtin_amdgcn_readlane(ib_incl, 63);
        exa[j] = ia_incl - sa;
        exb[j] = tot_a + ib_incl - sb2;
        if (lane == 0) wtot[j][wv] = tot_a + tot_b;
    }
    __syncthreads();
    int base = pre_s;
    const bool vec_ok = (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        int carry = base;
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            const int tt = wtot[j][w2];
            if (w2 < wv) carry += tt;
            base += tt;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int2 v = h ? vb[j] : va[j];
            const int2 ends = pair_at(en, j * 512 + wv * 128 + h * 64 + lane);
            const int g0 = carry + (h ? exb[j] : exa[j]) + v.x, g1 = g0 + v.y;
            const int i0 = j * 1024 + wv * 256 + h * 128 + 2 * lane;
            // numpy: ends / depth * 100 in float64, 0 where depth == 0 (frag/_cleavage_profile.py:208-210)
            const double o0 = g0 ? (double)ends.x / (double)g0 * 100.0 : 0.0;
            const double o1 = g1 ? (double)ends.y / (double)g1 * 100.0 : 0.0;
            if (i0 + 1 < len_t) {
                if (vec_ok) {
                    typedef double d2 __attribute__((ext_vector_type(2)));
                    d2 v2 = {o0, o1};
                    __builtin_nontemporal_store(v2, reinterpret_cast<d2*>(dst + i0));
                }
                else { dst[i0] = o0; dst[i0 + 1] = o1; }
            } else if (i0 < len_t) {
                dst[i0] = o0;
            }
        }
    }
}

// candidates of the bases [t0, t0 + len_t): fs < t0 + len_t and fe >= t0 (a - fragment ending exactly at t0 still
// puts an end there)
__device__ __forceinline__ int cleave_bound(const ContigView& cv, long long q, int which) {
    if (q <= 0) return 0;
    const long long kb = q >> kBinShift;
    return kb >= cv.n_bins ? cv.n : cv.bin_idx[kb + which];
}

__global__ __launch_bounds__(256) void cleavage_kernel(ContigView cv, CleaveParams p, const int64_t* iv_start_,
                                                       const int64_t* iv_stop_, const int64_t* out_off_,
                                                       const int32_t* tile_iv, const int32_t* tile_k,
                                                       double* __restrict__ out) {
    constexpr int T = kWpsTile, H = T / 2;
    // 16 KB for both layouts: 2 x T 16-bit counters, or (a tile with 32 768 candidates or more: ~900x depth,
    // chrM) 2 x T/2 32-bit counters for one half of the tile after the other
    __shared__ __attribute__((aligned(16))) int dd[H];
    __shared__ __attribute__((aligned(16))) int en[H];
    __shared__ int pre_s;
    __shared__ int wtot[T / 1024][4];
    const int tid = threadIdx.x;
    long long iv_start, iv_stop, out_off, k;
    if (tile_iv) {
        const int iv = tile_iv[blockIdx.x];
        iv_start = iv_start_[iv]; iv_stop = iv_stop_[iv]; out_off = out_off_[iv]; k = tile_k[blockIdx.x];
    } else {
        iv_start = p.start; iv_stop = p.stop; out_off = 0; k = blockIdx.x;
    }
    const long long t0 = iv_start + k * T;
    const int len_t = (int)(min(t0 + (long long)T, iv_stop) - t0);
    auto clear = [&]() {  // both arrays (16 KB whichever layout) and the carry word
        const int4 z = make_int4(0, 0, 0, 0);
        int4* a4 = reinterpret_cast<int4*>(dd);
        int4* b4 = reinterpret_cast<int4*>(en);
#pragma unroll
        for (int j = 0; j < H / 4 / 256; ++j) { a4[j * 256 + tid] = z; b4[j * 256 + tid] = z; }
        if (tid == 2) pre_s = 0;
    };
    // The candidates' bounds: two reads of the position index at block-uniform addresses, taken by every thread (scalar
    // loads, under way while the arrays are cleared; no hand-over through LDS, one barrier instead of two).
    const int lo = cleave_bound(cv, t0 - (long long)p.lmax, 0), hi = cleave_bound(cv, t0 + len_t, 1);
    clear();
    __syncthreads();
    double* dst = out + out_off + k * T;
    if (hi - lo < 32768) {
        cleave_tile<true, T>(cv, p, iv_start, iv_stop, t0, len_t, lo, hi, dst, dd, en, pre_s, wtot);
        return;
    }
    for (int half = 0; half < 2; ++half) {
        const long long th = t0 + (long long)half * H;
        const int len_h = min(len_t - half * H, H);
        if (len_h <= 0) break;
        __syncthreads();  // (the arrays of the first half are done with)
        const int lo_h = cleave_bound(cv, th - (long long)p.lmax, 0), hi_h = cleave_bound(cv, th + len_h, 1);
        clear();
        __syncthreads();
        cleave_tile<false, H>(cv, p, iv_start, iv_stop, th, len_h, lo_h, hi_h, dst + (long long)half * H, dd, en, pre_s, wtot);
    }
}

// ---------------------------------------------------------------------------
// DELFI per-bin GC count (frag/_delfi.py:476-490: ref_bases.count("G") + count("C") on the
// upper-cased window sequence) over a reference image resident in HBM: either the raw FASTA
// text of a contig (1 byte per base, line breaks included: ranges are byte offsets) or the
// packed DNA of a .2bit record (T=0 C=1 A=2 G=3, first base in the high bits: ranges are base
// positions, G/C = codes with the low bit set).  One wave per range, 16-byte loads.
// ---------------------------------------------------------------------------
__device__ __forceinline__ int gc_in_text_word(uint32_t w) {
    int c = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t b = ((w >> (8 * k)) & 0xffu) | 0x20u;  // fold case
        c += (b == 'g') | (b == 'c');
    }
    return c;
}

__global__ __launch_bounds__(256) void gc_count_kernel(const uint8_t* __restrict__ img, int64_t img_bytes, int kind,
                                                       const int64_t* lo_, const int64_t* hi_, int n,
                                                       int64_t* __restrict__ out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= n) return;
    int64_t lo = lo_[r], hi = hi_[r];
    long long acc = 0;
    if (hi > lo) {
        // byte range [b0, b1) of the image; for 2-bit, partial first/last bytes are masked
        const int64_t b0 = kind ? lo >> 2 : lo, b1 = kind ? (hi + 3) >> 2 : hi;
        const int64_t a0 = (b0 + 15) & ~15LL, a1 = b1 & ~15LL;  // 16-byte aligned interior
        auto byte_gc = [&](int64_t b) -> int {
            const uint32_t v = img[b];
            if (!kind) { const uint32_t x = v | 0x20u; return (x == 'g') | (x == 'c'); }
            int c = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int64_t pos = b * 4 + k;
                if (pos >= lo && pos < hi) c += (v >> (6 - 2 * k)) & 1u;
            }
            return c;
        };
        if (a0 >= a1) {
            for (int64_t b = b0 + lane; b < b1; b += 64) acc += byte_gc(b);
        } else {
            for (int64_t b = b0 + lane; b < a0; b += 64) acc += byte_gc(b);
            for (int64_t b = a1 + lane; b < b1; b += 64) acc += byte_gc(b);
            // interior bytes are whole: every base of a 2-bit byte is inside [lo, hi) unless it is the very
            // first or last byte of the range, which can only sit in the interior if it is complete
            const int64_t first_full = kind ? ((lo + 3) >> 2) : b0, last_full = kind ? (hi >> 2) : b1;
            auto chunk = [&](int64_t b, const uint4& v) {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                if (!kind) {
                    acc += gc_in_text_word(w[0]) + gc_in_text_word(w[1]) + gc_in_text_word(w[2]) + gc_in_text_word(w[3]);
                } else if (b >= first_full && b + 16 <= last_full) {
                    acc += __popc(w[0] & 0x55555555u) + __popc(w[1] & 0x55555555u) + __popc(w[2] & 0x55555555u) +
                           __popc(w[3] & 0x55555555u);
                } else {
                    for (int k = 0; k < 16; ++k) acc += byte_gc(b + k);
                }
            };
            // four 1 KB rows of the wave in flight per trip (a 100 kb bin of a 2bit image is 25 of them: the loop
            // is as long as its loads' latency, so the loads are issued together), then row by row.  Sixteen rows per
            // trip (two trips per bin) were measured: 16.7 us against 13.3 -- 2 432 waves with 16 KB each in flight are 39 MB
            // of requests at once, and HBM serves that many concurrent streams worse than fewer, longer ones
            int64_t b = a0 + 16 * lane;
            for (; b + 3 * 16 * 64 < a1; b += 4 * 16 * 64) {
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const uint4*>(img + b + u * 16 * 64);
#pragma unroll
                for (int u = 0; u < 4; ++u) chunk(b + u * 16 * 64, v[u]);
            }
            for (; b < a1; b += 16 * 64) chunk(b, *reinterpret_cast<const uint4*>(img + b));
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if (lane == 0) out[r] = acc;
}

// ---------------------------------------------------------------------------
// ordered selection of one window's fragments (frag_length / frag_array)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void select_count_kernel(ContigView cv, int lo, int hi, int ws, int we,
                                                           WinPred pred, uint32_t* block_cnt) {
    __shared__ int red[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lo + blockIdx.x * 256 + tid;
    int f = 0;
    if (i < hi) f = pred(cv, i, cv.start[i], cv.end[i], cv.mapq[i], ws, we, 0);
    int c = wave_reduce_add(f);
    if (lane == 0) red[wv] = c;
    __syncthreads();
    if (tid == 0) block_cnt[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void select_write_kernel(ContigView cv, int lo, int hi, int ws, int we,
                                                           WinPred pred, const uint32_t* block_off, int64_t cap,
                                                           int32_t* len_out, int32_t* start_out, int32_t* end_out,
                                                           uint8_t* mapq_out, uint8_t* strand_out,
                                                           int32_t* order_out) {
    __shared__ int wtot[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lo + blockIdx.x * 256 + tid;
    int f = 0, fs = 0, fe = 0, q = 0;
    if (i < hi) {
        fs = cv.start[i]; fe = cv.end[i]; q = cv.mapq[i];
        f = pred(cv, i, fs, fe, q, ws, we, 0);
    }
    unsigned long long m = __ballot(f);
    int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (lane == 0) wtot[wv] = __popcll(m);
    __syncthreads();
    int64_t off = block_off[blockIdx.x];
    for (int j = 0; j < wv; ++j) off += wtot[j];
    off += rank;
    if (f && off < cap) {
        if (len_out) len_out[off] = fe - fs;
        if (start_out) start_out[off] = fs;
        if (end_out) end_out[off] = fe;
        if (mapq_out) mapq_out[off] = (uint8_t)q;
        if (strand_out) strand_out[off] = cv.strand[i];
        if (order_out) order_out[off] = cv.order[i];
    }
}

__global__ void add_i64_kernel(const int64_t* a, const int64_t* b, int64_t* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = a[i] + b[i];
}

// ---------------------------------------------------------------------------
// Length statistics of every window from its dense histogram row (a9: frag/_frag_length.py:156-172 `_find_median`,
// :202-224 `_frag_length_stats`): one wavefront per window, the row read three times out of L2 (sums; squared
// deviations around the mean; the cumulative search of the median).  out[w] = mean, median, stdev, min, max, total,
// n_short as float64 (integers below 2^53: exact); a window without fragments gets zeros (total 0 says so).
//   mean   = sum(v c) / n            both exact integers, one IEEE division - Python's int / int
//   median = the reference's search: odd n looks for cdf >= n // 2 (not n // 2 + 1), even n averages the values at
//            cdf >= n // 2 and cdf >= n // 2 + 1; n // 2 == 0 (a single fragment) takes the first value
//   stdev  = sqrt(sum(c (v - mean)^2) / n), population; summed lane-wise and then across the wave (the reference sums
//            in dict insertion order: equal to ~1e-16 relative, the tests allow 1e-9)
// ---------------------------------------------------------------------------
__device__ __forceinline__ long long wave_sum_ll(long long v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

__global__ __launch_bounds__(256) void window_stats_kernel(const uint32_t* __restrict__ hist, int n_win, int n_bins,
                                                           int len_lo, int short_cut, double* __restrict__ out) {
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (w >= n_win) return;
    const int lane = threadIdx.x & 63;
    const uint32_t* h = hist + (size_t)w * (size_t)n_bins;
    long long tot = 0, sum = 0, n_short = 0;
    int first = INT32_MAX, last = -1;
    for (int b = lane; b < n_bins; b += 64) {
        const long long c = h[b];
        if (c) {
            const int v = len_lo + b;
            tot += c;
            sum += c * v;
            if (v <= short_cut) n_short += c;
            first = min(first, b);
            last = b;
        }
    }
    tot = wave_sum_ll(tot);
    sum = wave_sum_ll(sum);
    n_short = wave_sum_ll(n_short);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        first = min(first, __shfl_xor(first, d, 64));
        last = max(last, __shfl_xor(last, d, 64));
    }
    double* o = out + (size_t)w * 7;
    if (tot == 0) {
        if (lane < 7) o[lane] = 0.0;
        return;
    }
    const double mean = (double)sum / (double)tot;
    double var = 0.0;
    for (int b = lane; b < n_bins; b += 64) {
        const uint32_t c = h[b];
        if (c) {
            const double d = (double)(len_lo + b) - mean;
            var += (double)c * (d * d);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) var += __shfl_xor(var, d, 64);
    // the two cumulative searches, 64 bins a step
    const long long k1 = tot / 2;
    int i1 = k1 == 0 ? first : -1, i2 = -1;
    long long base = 0;
    for (int b0 = first & ~63; b0 < n_bins && (i1 < 0 || i2 < 0); b0 += 64) {
        long long s = b0 + lane < n_bins ? (long long)h[b0 + lane] : 0;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const long long t = __shfl_up(s, d, 64);
            if (lane >= d) s += t;
        }
        const long long cdf = base + s;
        const unsigned long long m1 = __ballot(cdf >= k1), m2 = __ballot(cdf >= k1 + 1);
        if (i1 < 0 && m1) i1 = b0 + __builtin_ctzll(m1);
        if (i2 < 0 && m2) i2 = b0 + __builtin_ctzll(m2);
        base += __shfl(s, 63, 64);
    }
    if (lane == 0) {
        o[0] = mean;
        o[1] = (tot & 1) ? (double)(len_lo + i1) : ((double)(len_lo + i1) + (double)(len_lo + i2)) / 2.0;
        o[2] = sqrt(var / (double)tot);
        o[3] = (double)(len_lo + first);
        o[4] = (double)(len_lo + last);
        o[5] = (double)tot;
        o[6] = (double)n_short;
    }
}

// ---------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------
void launch_add_i64(hipStream_t s, const int64_t* a, const int64_t* b, int64_t* out, int n) {
    hipLaunchKernelGGL(add_i64_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, out, n);
}

void launch_window_stats(hipStream_t s, const uint32_t* hist, int n_win, int n_bins, int len_lo, int short_cut, double* out) {
    if (n_win <= 0) return;
    hipLaunchKernelGGL(window_stats_kernel, dim3((n_win + 3) / 4), dim3(256), 0, s, hist, n_win, n_bins, len_lo, short_cut, out);
}

void launch_stats(hipStream_t s, const int32_t* start, const int32_t* end, int n, FragStats* st) {
    int blocks = min(1024, max(1, (n + kStatsThreads - 1) / kStatsThreads));
    hipLaunchKernelGGL(stats_kernel, dim3(blocks), dim3(kStatsThreads), 0, s, start, end, n, st);
}

void launch_bin_index(hipStream_t s, const int32_t* start, int n, int n_bins, int32_t* idx) {
    int threads = n_bins + 1;
    hipLaunchKernelGGL(bin_index_kernel, dim3((threads + 255) / 256), dim3(256), 0, s, start, n, n_bins, idx);
}

void launch_r1_inside(hipStream_t s, const int32_t* start, const int32_t* end, const int32_t* r1s, const int32_t* r1e,
                      int n, int* bad) {
    if (n <= 0) return;
    hipLaunchKernelGGL(r1_inside_kernel, dim3(min(2048, (n + 255) / 256)), dim3(256), 0, s, start, end, r1s, r1e, n, bad);
}

void launch_plan(hipStream_t s, const ContigView& cv, const int32_t* ws, const int32_t* we, int n_win, int lmax,
                 int small_max, const WindowPlan& pl, int64_t* const zero[4]) {
    ZeroList z;
    for (int k = 0; k < 4; ++k) z.p[k] = zero ? zero[k] : nullptr;
    if (n_win <= kPlanSingleBlockMax) {
        hipLaunchKernelGGL(plan_kernel, dim3(1), dim3(1024), 0, s, cv, ws, we, n_win, lmax, small_max, pl.cand_lo,
                           pl.cand_hi, pl.nchunks, pl.chunk_off, z);
        return;
    }
    hipLaunchKernelGGL(bounds_kernel, dim3((n_win + 255) / 256), dim3(256), 0, s, cv, ws, we, n_win, lmax, small_max,
                       pl.cand_lo, pl.cand_hi, pl.nchunks, z);
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, pl.nchunks, n_win, pl.chunk_off);
}

static WinPred make_win_pred(const ftk_filter& f) {
    return WinPred{f.mapq_min, f.min_len < 0 ? INT32_MIN : f.min_len, f.max_len < 0 ? INT32_MAX : f.max_len, f.policy,
                   f.fetch_mode == FTK_FETCH_BAM_READ1};
}

// The FAST block kernels serve the common request: midpoint policy, no length bounds on the coverage /
// histogram filter, and (when coverage and DELFI run together) one mapq cut for both.  FTK_FEAT_FAST=0
// keeps the general kernels (tests run both).
static bool feat_fast_ok(const FeatParams& P, bool ch, bool df) {
    static const int allow = getenv("FTK_FEAT_FAST") ? atoi(getenv("FTK_FEAT_FAST")) : 1;
    if (!allow) return false;
    if (ch && (P.is_any || P.ch_min > 0 || P.ch_max < (1 << 30))) return false;
    if (ch && df && P.ch_q != P.df_q) return false;
    return true;
}

template <int CH, bool DF, bool BAM>
static bool launch_feat_t(hipStream_t s, int grid_large, const ContigView& cv, const int32_t* ws, const int32_t* we,
                          int n_win, const WindowPlan& pl, const FeatParams& P, bool small_path, int block_lmax,
                          int block_threads, const WpsTail* tail) {
    const size_t lds1 = (CH && P.do_hist) ? (size_t)(P.n_bins + 1) * sizeof(uint32_t) : 0;  // + overflow bin
    if (tail && tail->n_tiles > 0 && block_lmax >= 0 && CH < 2 && feat_fast_ok(P, CH != 0, DF) &&
        (long long)n_win + tail->n_tiles < (1LL << 31)) {
        // the merged launch: feature blocks first, the WPS tiles behind them (feat_then_wps_kernel)
        FeatParams Pf = P;
        if (!CH) Pf.ch_q = P.df_q;
        const dim3 grid((unsigned)((long long)n_win + tail->n_tiles));
#define FTK_MERGED(HIST, NT)                                                                                         \
    hipLaunchKernelGGL((feat_then_wps_kernel<CH != 0, HIST, DF, BAM, NT>), grid, dim3(256), lds1, s, cv, ws, we, n_win, \
                       block_lmax, Pf, tail->p, (long long)tail->n_tiles, tail->out)
        if (CH && P.do_hist) { if (tail->p.nt_store) FTK_MERGED(true, true); else FTK_MERGED(true, false); }
        else { if (tail->p.nt_store) FTK_MERGED(false, true); else FTK_MERGED(false, false); }
#undef FTK_MERGED
        return true;
    }
    if (block_lmax >= 0 && CH < 2 && feat_fast_ok(P, CH != 0, DF)) {
#define FTK_FAST(BS)                                                                                                \
    do {                                                                                                            \
        if (CH && P.do_hist)                                                                                        \
            hipLaunchKernelGGL((feat_fast_kernel<BS, CH != 0, true, DF, BAM>), dim3(n_win), dim3(BS), lds1, s, cv, ws, we,   \
                               n_win, block_lmax, Pf);                                                              \
        else                                                                                                        \
            hipLaunchKernelGGL((feat_fast_kernel<BS, CH != 0, false, DF, BAM>), dim3(n_win), dim3(BS), lds1, s, cv, ws, we,  \
                               n_win, block_lmax, Pf);                                                              \
    } while (0)
        FeatParams Pf = P;
        if (!CH) Pf.ch_q = P.df_q;  // the shared mapq term
        if (block_threads >= 512) FTK_FAST(512);
        else FTK_FAST(256);
#undef FTK_FAST
        return false;
    }
    if (block_lmax >= 0) {
        if (block_threads >= 512)
            hipLaunchKernelGGL((feat_block_kernel<512, CH, DF, BAM>), dim3(n_win), dim3(512), lds1, s, cv, ws, we, n_win,
                               block_lmax, P);
        else
            hipLaunchKernelGGL((feat_block_kernel<256, CH, DF, BAM>), dim3(n_win), dim3(256), lds1, s, cv, ws, we, n_win,
                               block_lmax, P);
        return false;
    }
    if (small_path)
        hipLaunchKernelGGL((feat_small_kernel<CH, DF, BAM>), dim3((n_win + 3) / 4), dim3(256), 4 * lds1, s, cv, ws, we,
                           n_win, pl.cand_lo, pl.cand_hi, pl.nchunks, P);
    // the motif stream keeps four waves per SIMD busy with a quarter of the blocks: fewer window searches and
    // histogram flushes per fragment (110 -> 101 us for a chr2-sized contig in 1 Mb windows; 16 / 6 per CU: 103 / 112)
    const int grid = CH == kMotifWord ? std::max(grid_large / 4, 1) : grid_large;
    hipLaunchKernelGGL((feat_large_kernel<CH, DF, BAM>), dim3(grid), dim3(256), lds1, s, cv, ws, we, n_win,
                       pl.cand_lo, pl.cand_hi, pl.chunk_off, P);
    return false;
}

// One pass computing any combination of {coverage, length histogram} (filter `f`) and DELFI
// short/long.  With small_path the wave-per-window kernel writes or clears every histogram row;
// without it hist_out / over_out must be zero-filled by the caller.
// `tail`: a whole-interval WPS launch of the same contig that should run in the SAME launch behind the feature blocks
// (taken on the FAST block path only); returns true when it did, false when the caller must launch it itself.
bool launch_window_features(hipStream_t s, int grid_large, const ContigView& cv, const int32_t* ws, const int32_t* we,
                            int n_win, const WindowPlan& pl, const FeatureRequest& r, bool small_path, int block_lmax,
                            const WpsTail* tail) {
    FeatParams P{};
    const bool ch = r.cov_out || r.hist_out;
    const bool df = r.short_out != nullptr;
    if (r.filter) P.wp = make_win_pred(*r.filter);
    P.do_cov = r.cov_out != nullptr;
    P.do_hist = r.hist_out != nullptr;
    P.len_lo = r.len_lo;
    P.n_bins = r.n_bins;
    P.cov_out = r.cov_out;
    P.hist_out = r.hist_out;
    P.over_out = r.over_out;
    P.short_out = r.short_out;
    P.long_out = r.long_out;
    {
        int gc[4];
        gap_constants(r.gaps, gc);
        P.cen0 = gc[0]; P.cen1 = gc[1]; P.tel0 = gc[2]; P.tel1 = gc[3];
        P.dp = DelfiPred{r.delfi_mapq_min, gc[0], gc[1], gc[2], gc[3], r.bl_off, r.bl_r0, r.bl_pm};
        P.df_q = std::min(std::max(r.delfi_mapq_min, 0), 256);
    }
    if (r.filter) {
        const ftk_filter& f = *r.filter;
        P.ch_q = std::min(std::max(f.mapq_min, 0), 256);
        P.ch_min = f.min_len < 0 ? 0 : std::min(f.min_len, 1 << 30);
        P.ch_max = f.max_len < 0 ? (1 << 30) : std::min(f.max_len, 1 << 30);
        P.is_any = f.policy == FTK_POLICY_ANY;
    }
    const bool bam = cv.r1_start != nullptr && (!r.filter || r.filter->fetch_mode == FTK_FETCH_BAM_READ1);
#define FTK_FEAT(CH, DF)                                                                              \
    do {                                                                                              \
        if (bam) merged = launch_feat_t<CH, DF, true>(s, grid_large, cv, ws, we, n_win, pl, P, small_path, block_lmax, r.block_threads, tail);  \
        else merged = launch_feat_t<CH, DF, false>(s, grid_large, cv, ws, we, n_win, pl, P, small_path, block_lmax, r.block_threads, tail);     \
    } while (0)
    bool merged = false;
    if (r.motif) {
        P.mp = *r.motif;
        P.do_hist = 1;
        // a 2bit image and k <= 13: the k-mer is cut out of one 4-byte load (its own instantiation: the general form's
        // loops, unrolled with the window kernels, do not fit the instruction cache beside it)
        if (P.mp.kind == FTK_REF_2BIT && P.mp.k <= 13 && P.is_any && P.ch_min <= 0 && P.ch_max >= (1 << 30)) FTK_FEAT(kMotifWord, false);
        else FTK_FEAT(2, false);
    } else if (ch && df) FTK_FEAT(1, true);
    else if (ch) FTK_FEAT(1, false);
    else if (df) FTK_FEAT(0, true);
#undef FTK_FEAT
    return merged;
}

void gap_constants(const ftk_gaps& g, int out[4]) {
    int cen0 = INT32_MAX, cen1 = INT32_MIN, tel0 = INT32_MAX, tel1 = INT32_MIN;
    if (g.has_gaps) {
        cen0 = g.cen_start;
        cen1 = g.cen_stop;
        if (g.n_telo > 0) {
            tel0 = INT32_MIN;
            tel1 = INT32_MAX;
            for (int t = 0; t < g.n_telo; ++t) {
                tel0 = std::max(tel0, g.telo_start[t]);
                tel1 = std::min(tel1, g.telo_stop[t]);
            }
        }
    }
    auto clamp_c = [](int v) { return std::min(std::max(v, -1), (1 << 30) + 1); };
    out[0] = clamp_c(cen0); out[1] = clamp_c(cen1); out[2] = clamp_c(tel0); out[3] = clamp_c(tel1);
}

void launch_window_features_batch(hipStream_t s, const FeatItem* d_items, int n_items, int total_win,
                                  const FeatureRequest& r, bool bam) {
    FeatParams P{};
    const bool ch = r.cov_out || r.hist_out;
    const bool df = r.short_out != nullptr;
    if (r.filter) {
        const ftk_filter& f = *r.filter;
        P.wp = make_win_pred(f);
        P.ch_q = std::min(std::max(f.mapq_min, 0), 256);
        P.ch_min = f.min_len < 0 ? 0 : std::min(f.min_len, 1 << 30);
        P.ch_max = f.max_len < 0 ? (1 << 30) : std::min(f.max_len, 1 << 30);
        P.is_any = f.policy == FTK_POLICY_ANY;
    }
    P.df_q = std::min(std::max(r.delfi_mapq_min, 0), 256);
    P.do_cov = r.cov_out != nullptr;
    P.do_hist = r.hist_out != nullptr;
    P.len_lo = r.len_lo;
    P.n_bins = r.n_bins;
    P.cov_out = r.cov_out;
    P.hist_out = r.hist_out;
    P.over_out = r.over_out;
    P.short_out = r.short_out;
    P.long_out = r.long_out;
    const size_t lds1 = P.do_hist ? (size_t)(P.n_bins + 1) * sizeof(uint32_t) : 0;
    if (feat_fast_ok(P, ch, df)) {
        FeatParams Pf = P;
        if (!ch) Pf.ch_q = P.df_q;
#define FTK_FASTB(CHK, HIST, DF)                                                                                   \
    do {                                                                                                           \
        if (bam) hipLaunchKernelGGL((feat_fast_batch_kernel<512, CHK, HIST, DF, true>), dim3(total_win), dim3(512), lds1, s, \
                                    d_items, n_items, Pf);                                                         \
        else hipLaunchKernelGGL((feat_fast_batch_kernel<512, CHK, HIST, DF, false>), dim3(total_win), dim3(512), lds1, s,    \
                                d_items, n_items, Pf);                                                             \
    } while (0)
        if (ch && df) { if (P.do_hist) FTK_FASTB(true, true, true); else FTK_FASTB(true, false, true); }
        else if (ch) { if (P.do_hist) FTK_FASTB(true, true, false); else FTK_FASTB(true, false, false); }
        else if (df) FTK_FASTB(false, false, true);
#undef FTK_FASTB
        return;
    }
#define FTK_FEATB(CH, DF)                                                                                           \
    do {                                                                                                            \
        if (bam) hipLaunchKernelGGL((feat_batch_kernel<512, CH, DF, true>), dim3(total_win), dim3(512), lds1, s, d_items, \
                                    n_items, P);                                                                    \
        else hipLaunchKernelGGL((feat_batch_kernel<512, CH, DF, false>), dim3(total_win), dim3(512), lds1, s, d_items,   \
                                n_items, P);                                                                        \
    } while (0)
    if (ch && df) FTK_FEATB(1, true);
    else if (ch) FTK_FEATB(1, false);
    else if (df) FTK_FEATB(0, true);
#undef FTK_FEATB
}

void launch_wps(hipStream_t s, const ContigView& cv, const WpsParams& p, int64_t n_tiles, const int64_t* iv_start,
                const int64_t* iv_stop, const int64_t* out_off, const int32_t* tile_iv, const int32_t* tile_k,
                int64_t* out) {
    if (n_tiles <= 0) return;
    // tiles per block: with non-temporal score stores one tile per block is fastest (6.4 TB/s vs 6.2 at
    // 2, 5.9-6.2 at 3-8 on chr2); with ordinary stores 2 was.  FTK_WPS_TPB overrides for experiments.
    static const long long tpb_env = getenv("FTK_WPS_TPB") ? atoll(getenv("FTK_WPS_TPB")) : 0;
    const long long tpb = tpb_env > 0 ? tpb_env : 1;
    const long long grid = (n_tiles + tpb - 1) / tpb;
#define FTK_WPS(MULTI, NT)                                                                                              \
    hipLaunchKernelGGL((wps_stream_kernel<MULTI, false, false, NT>), dim3((unsigned)grid), dim3(256), 0, s, cv, p, iv_start, \
                       iv_stop, out_off, tile_iv, tile_k, (long long)n_tiles, (int)tpb, out, (const WpsItem*)nullptr, 0,    \
                       FusedParams{})
    if (tpb == 1) { if (p.nt_store) FTK_WPS(false, true); else FTK_WPS(false, false); }
    else { if (p.nt_store) FTK_WPS(true, true); else FTK_WPS(true, false); }
#undef FTK_WPS
}

// Whole-interval WPS with the window features of a regular bin tiling in the same pass.
void launch_wps_fused(hipStream_t s, const ContigView& cv, const WpsParams& p, int64_t n_tiles, const FusedParams& F,
                      int64_t* out) {
    if (n_tiles <= 0) return;
    const size_t lds = F.do_hist ? (size_t)(F.n_bins + 1) * 4 : 0;
#define FTK_WPSF(NT)                                                                                               \
    hipLaunchKernelGGL((wps_stream_kernel<false, false, true, NT>), dim3((unsigned)n_tiles), dim3(256), lds, s, cv, p, \
                       (const int64_t*)nullptr, (const int64_t*)nullptr, (const int64_t*)nullptr,                      \
                       (const int32_t*)nullptr, (const int32_t*)nullptr, (long long)n_tiles, 1, out,                   \
                       (const WpsItem*)nullptr, 0, F)
    if (p.nt_store) FTK_WPSF(true); else FTK_WPSF(false);
#undef FTK_WPSF
}

// Several (contig, interval) items in one launch, one 4096-base tile per block.
void launch_wps_batch(hipStream_t s, const WpsParams& p, const WpsItem* d_items, int n_items, int64_t n_tiles,
                      int64_t* out) {
    if (n_tiles <= 0) return;
    ContigView none{};
#define FTK_WPSB(NT)                                                                                                \
    hipLaunchKernelGGL((wps_stream_kernel<false, true, false, NT>), dim3((unsigned)n_tiles), dim3(256), 0, s, none, p,  \
                       (const int64_t*)nullptr, (const int64_t*)nullptr, (const int64_t*)nullptr,                       \
                       (const int32_t*)nullptr, (const int32_t*)nullptr, (long long)n_tiles, 1, out, d_items, n_items,  \
                       FusedParams{})
    if (p.nt_store) FTK_WPSB(true); else FTK_WPSB(false);
#undef FTK_WPSB
}

void launch_cleavage(hipStream_t s, const ContigView& cv, const CleaveParams& p, int64_t n_tiles,
                     const int64_t* iv_start, const int64_t* iv_stop, const int64_t* out_off, const int32_t* tile_iv,
                     const int32_t* tile_k, double* out) {
    if (n_tiles <= 0) return;
    hipLaunchKernelGGL(cleavage_kernel, dim3((unsigned)n_tiles), dim3(256), 0, s, cv, p, iv_start, iv_stop, out_off,
                       tile_iv, tile_k, out);
}

void launch_gc_count(hipStream_t s, const uint8_t* img, int64_t img_bytes, int kind, const int64_t* lo,
                     const int64_t* hi, int n, int64_t* out) {
    if (n <= 0) return;
    hipLaunchKernelGGL(gc_count_kernel, dim3((n + 3) / 4), dim3(256), 0, s, img, img_bytes, kind, lo, hi, n, out);
}

void launch_select_count(hipStream_t s, const ContigView& cv, int lo, int hi, int ws, int we, const ftk_filter& f,
                         uint32_t* block_cnt) {
    WinPred pred = make_win_pred(f);
    int nb = (hi - lo + 255) / 256;
    if (nb <= 0) return;
    hipLaunchKernelGGL(select_count_kernel, dim3(nb), dim3(256), 0, s, cv, lo, hi, ws, we, pred, block_cnt);
}

void launch_scan_u32(hipStream_t s, const uint32_t* in, int n, uint32_t* off) {
    hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, s, in, n, off);
}

void launch_select_write(hipStream_t s, const ContigView& cv, int lo, int hi, int ws, int we, const ftk_filter& f,
                         const uint32_t* block_off, int64_t cap, int32_t* len_out, int32_t* start_out,
                         int32_t* end_out, uint8_t* mapq_out, uint8_t* strand_out, int32_t* order_out) {
    WinPred pred = make_win_pred(f);
    int nb = (hi - lo + 255) / 256;
    if (nb <= 0) return;
    hipLaunchKernelGGL(select_write_kernel, dim3(nb), dim3(256), 0, s, cv, lo, hi, ws, we, pred, block_off, cap,
                       len_out, start_out, end_out, mapq_out, strand_out, order_out);
}

}  // namespace ftk
