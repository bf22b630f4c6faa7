// WPS post-processing kernels: running median / mean subtraction and the
// Savitzky-Golay pass (reference: frag/_adjust_wps.py:25-50, 119-140).
//
// Running median of a W-wide window, one tile of outputs per block:
//   1. the tile's n_out + W - 1 inputs are turned into order-preserving 64-bit
//      keys and bitonic-sorted in LDS together with their positions;
//   2. each output walks the sorted slots in ascending order and counts the
//      slots whose position falls inside its window [o, o + W); the slot at
//      which the count reaches W/2 is the lower middle element, the next member
//      the upper one (ties need no special care: slots are distinct).
// All lanes of a wave read the same slots in step 2 (LDS broadcast reads), and a
// wave stops as soon as all of its outputs are resolved: about n_in / 2 slots
// per output instead of W reads per bit of a radix descent.
//
// Round 6: raw WPS scores are small INTEGERS (a bigWig of `multi_wps` holds counts), and the median of integers in a
// narrow range needs no sort: `adjust_median_hist_kernel` gives every lane a run of consecutive outputs and a
// histogram of its own (128 or 256 bins of 16 bits in LDS, two lanes to a word); the window slides by one - one bin
// down, one up - and the two middle order statistics follow by O(1) steps from where they were.  A tile whose values
// are not integers within a range of 256 marks its interval, and the sort kernel redoes the marked intervals: same
// result, bit for bit (the medians are the same input values).  10 000 x 5 kb runs at W = 1000: 7.2 ms -> 0.52 ms.
#include "ftk_kernels.h"

#include <algorithm>
#include <cmath>

namespace ftk {

namespace {

constexpr int kAdjThreads = 256;

__device__ __forceinline__ unsigned long long f64_key(double v) {
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ULL);
}
__device__ __forceinline__ double key_f64(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7fffffffffffffffULL) : ~k;
    return __longlong_as_double((long long)b);
}

// median_window W (even), n_sort = power of two >= n_out + W - 1
__device__ __forceinline__ void adjust_median_tile(const double* __restrict__ scores, const AdjustTile& t,
                                                   const double* __restrict__ edge_sub, int W, int n_sort,
                                                   double* __restrict__ out, const int* __restrict__ todo,
                                                   unsigned long long* lds_keys) {
    unsigned int* pos = (unsigned int*)(lds_keys + n_sort);  // [n_sort] input position of each sorted slot
    if (todo && todo[t.interval] != 2) return;  // a histogram kernel has answered this interval
    const double sub = edge_sub ? edge_sub[t.interval] : 0.0;
    const double* in = scores + t.in_base;
    const int n_in = t.n_out + W - 1;
    for (int i = threadIdx.x; i < n_sort; i += kAdjThreads) {
        lds_keys[i] = i < n_in ? f64_key(in[i] - sub) : ~0ULL;
        pos[i] = (unsigned int)i;
    }
    __syncthreads();
    for (int k = 2; k <= n_sort; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = threadIdx.x; i < n_sort / 2; i += kAdjThreads) {
                const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int b = a | j;
                const bool up = (a & k) == 0;
                const unsigned long long ka = lds_keys[a], kb = lds_keys[b];
                if ((ka > kb) == up) {
                    lds_keys[a] = kb;
                    lds_keys[b] = ka;
                    const unsigned int pa = pos[a];
                    pos[a] = pos[b];
                    pos[b] = pa;
                }
            }
            __syncthreads();
        }
    }
    // Selection: walk the sorted slots in order, counting those whose position lies in this
    // output's window; the slot where the count reaches W/2 holds the lower middle value, the
    // next member the upper one.  Every lane reads the same slots (LDS broadcast, 4 per read).
    const unsigned int uW = (unsigned int)W;
    const int target = W / 2;
    const uint4* pos4 = (const uint4*)pos;
    const int n_blk = n_sort / 4;
    for (int o0 = 0; o0 < t.n_out; o0 += kAdjThreads) {
        const int o = o0 + (int)threadIdx.x;
        const bool live = o < t.n_out;
        const unsigned int uo = (unsigned int)o;
        int cnt = 0, blk = -1, cnt_at = 0;
        for (int b = 0; b < n_blk; ++b) {
            const uint4 p = pos4[b];
            const int before = cnt;
            cnt += (p.x - uo < uW) + (p.y - uo < uW) + (p.z - uo < uW) + (p.w - uo < uW);
            if (blk < 0 && cnt >= target) {
                blk = b;
                cnt_at = before;
            }
            if (__all(blk >= 0 || !live)) break;
        }
        if (!live) continue;
        int s = blk * 4, c = cnt_at, s1 = -1;
        for (;; ++s) {
            if (pos[s] - uo < uW) {
                ++c;
                if (c == target) s1 = s;
                else if (c > target) break;
            }
        }
        const double med = (key_f64(lds_keys[s1]) + key_f64(lds_keys[s])) * 0.5;
        out[t.out_base + o] = (in[o + W / 2] - sub) - med;
    }
}

// With histogram kernels in front (left != NULL) a block leaves at once when they marked nothing at all (left[1] == 0):
// 40 000 blocks that each load their tile and look up its interval's mark were 18 us of an 0.5 ms call; leaving on
// left[1] alone they are 9.  (Smaller grids striding over the tiles were measured for the rest: 1 536 blocks - one per
// resident slot - cost the sorting case 21 %, 8 000 blocks of five tiles each 8 %, for 5 us of the other.)
__global__ __launch_bounds__(kAdjThreads) void adjust_median_kernel(const double* __restrict__ scores,
                                                                     const AdjustTile* __restrict__ tiles, int n_tiles,
                                                                     const double* __restrict__ edge_sub, int W,
                                                                     int n_sort, double* __restrict__ out,
                                                                     const int* __restrict__ todo,
                                                                     const int* __restrict__ left) {
    extern __shared__ unsigned long long lds_keys[];       // [n_sort] sorted keys, [n_sort] positions
    if (left && left[1] == 0) return;
    for (int ti = blockIdx.x; ti < n_tiles; ti += gridDim.x) {
        adjust_median_tile(scores, tiles[ti], edge_sub, W, n_sort, out, todo, lds_keys);
        __syncthreads();
    }
}

// ---- the median of small integers: one lane, one run of outputs, one sliding histogram ---------------------------
// Two sizes: 128 bins (16 KB of histograms per wavefront: seven tiles per CU - the ALU and LDS latencies of one lane's
// chain are hidden by the other waves of its SIMD; raw WPS at 30x spans 60-80 values over 5 kb) and 256 bins (32 KB,
// four per CU) for the tiles the first leaves (todo == kTodoWide), which in turn leaves the rest to the sort kernel
// (todo == kTodoSort).  An interval is redone as a whole by the next kernel of the chain when one of its tiles asks.
constexpr int kFastThreads = 64;    // one wavefront per tile
constexpr int kTodoWide = 1, kTodoSort = 2;

template <int kFastBins>
__device__ __forceinline__ void adjust_median_hist_tile(const double* __restrict__ scores, const AdjustTile& t,
                                                        const double* __restrict__ edge_sub, int W,
                                                        double* __restrict__ out, int* __restrict__ todo,
                                                        int* __restrict__ left, unsigned int* lds_fast) {
    // (the inputs are parked as int16 offsets where the histograms go: 128 bins x 64 lanes x 2 B hold the largest tile)
    constexpr int kHistWords = kFastBins / 2 * 64;
    static_assert((kAdjustFastTile + kAdjustMaxWindow) * 2 <= kHistWords * 4, "the parked inputs must fit where the histograms go");
    unsigned int* hh = lds_fast;                                   // [kFastBins][32]: bin b of lanes 2p (low half), 2p + 1 (high half)
    unsigned char* sv = (unsigned char*)(lds_fast + kHistWords);   // [n_in]: value - base
    if (kFastBins > 128 && todo[t.interval] != kTodoWide) return;  // (answered by the 128-bin pass)
    const int lane = threadIdx.x;
    const double* in = scores + t.in_base;
    const int n_in = t.n_out + W - 1;
    // integers (no -0.0) within a range of kFastBins?  The tile's inputs are read ONCE and parked as int16 where the
    // histograms will be; then shifted to bytes, then the histograms cleared.
    short* parked = (short*)hh;  // value - the tile's first value (a tile the histograms can take spans < 256)
    const double v0 = in[0];
    double lo = INFINITY, hi = -INFINITY;
    bool ok = true;
    // kParkRows rows of the wave in flight per trip.  The histograms leave room for seven of these one-wave blocks per
    // CU, so what a CU has in flight is what its seven waves ask for at once: with 8 rows (28 KB per CU) this loop
    // alone was 240 of the kernel's 524 us -- 400 MB read at 1.7 TB/s.
    constexpr int kParkRows = 32;
    for (int i0 = 0; i0 < n_in; i0 += kParkRows * kFastThreads) {
        double v[kParkRows];
#pragma unroll
        for (int u = 0; u < kParkRows; ++u) {
            const int i = i0 + u * kFastThreads + lane;
            v[u] = in[min(i, n_in - 1)];  // (no load under a branch)
        }
#pragma unroll
        for (int u = 0; u < kParkRows; ++u) {
            const int i = i0 + u * kFastThreads + lane;
            if (i < n_in) {
                const double d = v[u] - v0;
                ok = ok && v[u] == rint(v[u]) && fabs(v[u]) < 1e9 && fabs(d) < 30000.0 && !(v[u] == 0.0 && signbit(v[u]));
                lo = fmin(lo, v[u]);
                hi = fmax(hi, v[u]);
                parked[i] = (short)(int)d;
            }
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        lo = fmin(lo, __shfl_xor(lo, d, 64));
        hi = fmax(hi, __shfl_xor(hi, d, 64));
    }
    const bool all_ok = __all(ok);
    if (!all_ok || !(hi - lo < (double)kFastBins)) {
        // (racing writers of one interval: the larger request must win)
        if (lane == 0) {
            const int want = (all_ok && kFastBins <= 128 && hi - lo < 256.0) ? kTodoWide : kTodoSort;
            atomicMax(&todo[t.interval], want);
            left[want == kTodoWide ? 0 : 1] = 1;  // (racing writers all write 1) something is left for the next kernels
        }
        return;
    }
    const int base = (int)lo, rel = (int)(lo - v0);
    __syncthreads();
    {   // four values a step: two dwords of int16 in, one dword of bytes out (the parked array and `sv` are 4-byte
        // aligned; the last step may convert up to three slots past n_in, which both arrays have)
        const uint2* p4 = reinterpret_cast<const uint2*>(parked);
        unsigned int* s4 = reinterpret_cast<unsigned int*>(sv);
        for (int i = lane; i < (n_in + 3) / 4; i += kFastThreads) {
            const uint2 v = p4[i];
            const int a = (short)(v.x & 0xffffu) - rel, b = (short)(v.x >> 16) - rel, c = (short)(v.y & 0xffffu) - rel,
                      d = (short)(v.y >> 16) - rel;
            s4[i] = (unsigned)(a & 0xff) | ((unsigned)(b & 0xff) << 8) | ((unsigned)(c & 0xff) << 16) | ((unsigned)(d & 0xff) << 24);
        }
    }
    __syncthreads();
    {
        uint4* h4 = reinterpret_cast<uint4*>(hh);
        for (int k = lane; k < kFastBins / 2 * 64 / 4; k += kFastThreads) h4[k] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    // A lane's run of outputs starts on a multiple of 4 bytes of `sv` (it reads its values four at a time), and the
    // runs' length in dwords is ODD: lane l then starts in bank (l x odd) mod 32 - all 32 banks, two lanes each, the
    // least a wavefront can have.  (With the obvious 64 outputs per lane every lane read the same two banks: 0.93 ms
    // per 10 000 x 5 kb intervals instead of what follows.)
    int L = ((t.n_out + kFastThreads - 1) / kFastThreads + 3) & ~3;
    if (!((L >> 2) & 1)) L += 4;
    const int o_begin = lane * L, o_end = min(o_begin + L, t.n_out);
    if (o_begin >= t.n_out) return;
    const double sub = edge_sub ? edge_sub[t.interval] : 0.0;
    // bin b of lanes 2p and 2p + 1 share the word hh[b * 32 + p] (low / high half): a value's bin is its word's row, the
    // half and the increment are the lane's own constants - two instructions of address arithmetic per value
    unsigned int* mine = hh + (lane >> 1);
    const int half = (lane & 1) * 16;
    const unsigned int one = 1u << half;
    auto add = [&](int b) { atomicAdd(&mine[b * 32], one); };   // (no carry: a bin holds <= W <= 2048)
    auto take = [&](int b) { atomicSub(&mine[b * 32], one); };  // (no borrow: the bin held the value)
    auto cnt = [&](int b) { return (int)((mine[b * 32] >> half) & 0xffffu); };
    {
        int j = 0;
        for (; j + 16 <= W; j += 16) {  // (the adds return nothing: sixteen of them queue up behind four 4-byte reads)
            unsigned int pk[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) pk[r] = *reinterpret_cast<const unsigned int*>(sv + o_begin + j + 4 * r);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int u = 0; u < 4; ++u) add((pk[r] >> (8 * u)) & 0xffu);
        }
        for (; j + 4 <= W; j += 4) {
            const unsigned int pk = *reinterpret_cast<const unsigned int*>(sv + o_begin + j);
#pragma unroll
            for (int u = 0; u < 4; ++u) add((pk >> (8 * u)) & 0xffu);
        }
        for (; j < W; ++j) add(sv[o_begin + j]);
    }
    const int tgt = W / 2;  // the lower middle is the tgt-th smallest, the upper one the next
    // m: the bin of the lower middle; below: values in bins < m; cm: values in bin m.  below < tgt <= below + cm.
    int m = 0, below = 0, cm = cnt(0);
    while (below + cm < tgt) {
        below += cm;
        ++m;
        cm = cnt(m);
    }
    // The run, four outputs per trip: the three byte streams a step reads -- the value leaving the window, the one
    // entering it, the one at its centre -- are consecutive in `sv`, so one trip's twelve bytes are three (five, when
    // W / 2 or W is not a multiple of 4) dword reads issued together and waited for once.  (One byte read at a time
    // every step began with an LDS round trip behind the queue of atomics: 247 of the kernel's 502 us.)
    const int n_steps = o_end - o_begin;
    const unsigned char* p_old = sv + o_begin;             // 4-byte aligned
    const unsigned char* p_new = sv + o_begin + W;         // W & 3 into a dword (uniform)
    const unsigned char* p_cen = sv + o_begin + W / 2;
    const int sh_new = W & 3, sh_cen = (W / 2) & 3;
    auto four = [](const unsigned char* p, int sh) -> unsigned int {
        const unsigned int* q = reinterpret_cast<const unsigned int*>(p - sh);
        const unsigned int lo = q[0];
        if (sh == 0) return lo;
        return __builtin_amdgcn_alignbyte(q[1], lo, (unsigned)sh);
    };
    for (int s0 = 0; s0 < n_steps; s0 += 4) {
        const unsigned int old4 = *reinterpret_cast<const unsigned int*>(p_old + s0);
        const unsigned int new4 = four(p_new + s0, sh_new);
        const unsigned int cen4 = four(p_cen + s0, sh_cen);
        double r[4];  // the trip's outputs: written as 2 x 16 bytes (a lane's run is its own stretch of `out`: one 8-byte
                      // store per step made every store instruction 64 partial sectors -- 2.1 x the output in HBM writes)
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int st = s0 + u;
            if (st >= n_steps) break;
            int m2 = m;
            if (below + cm < tgt + 1) {  // (rare with hundreds of values in a few dozen bins)
                m2 = m + 1;
                while (cnt(m2) == 0) ++m2;
            }
            // the sort kernel's arithmetic on the same values: ((x1 - sub) + (x2 - sub)) * 0.5, x - sub - median
            const double med = (((double)(base + m) - sub) + ((double)(base + m2) - sub)) * 0.5;
            r[u] = ((double)(base + (int)((cen4 >> (8 * u)) & 0xffu)) - sub) - med;
            if (u == 3 || st + 1 >= n_steps) {  // the trip's (or the run's) last output: hand the trip over
                double* dst = out + t.out_base + o_begin + s0;
                if (u == 3) {
                    typedef double __attribute__((ext_vector_type(2), aligned(8))) d2u;
                    *reinterpret_cast<d2u*>(dst) = d2u{r[0], r[1]};
                    *reinterpret_cast<d2u*>(dst + 2) = d2u{r[2], r[3]};
                } else {
                    dst[0] = r[0];
                    if (u >= 1) dst[1] = r[1];
                    if (u >= 2) dst[2] = r[2];
                }
            }
            if (st + 1 >= n_steps) break;
            const int b_old = (int)((old4 >> (8 * u)) & 0xffu), b_new = (int)((new4 >> (8 * u)) & 0xffu);
            if (b_old == b_new) continue;
            take(b_old);
            add(b_new);
            below += (b_new < m) - (b_old < m);
            cm += (b_new == m) - (b_old == m);
            // the middle moves by a bin or two at most: the counts of the bins it crosses are the only reads of a step
            while (below >= tgt) {
                --m;
                cm = cnt(m);
                below -= cm;
            }
            while (below + cm < tgt) {
                below += cm;
                ++m;
                cm = cnt(m);
            }
        }
    }
}

// The 128-bin pass has a block per tile; the 256-bin pass behind it strides over the tiles with a small grid and leaves
// at once when the first marked nothing for it (left[0] == 0).
template <int kFastBins>
__global__ __launch_bounds__(kFastThreads) void adjust_median_hist_kernel(const double* __restrict__ scores,
                                                                           const AdjustTile* __restrict__ tiles, int n_tiles,
                                                                           const double* __restrict__ edge_sub, int W,
                                                                           double* __restrict__ out, int* __restrict__ todo,
                                                                           int* __restrict__ left) {
    extern __shared__ unsigned int lds_fast[];
    if (kFastBins > 128 && left[0] == 0) return;
    for (int ti = blockIdx.x; ti < n_tiles; ti += gridDim.x) {
        adjust_median_hist_tile<kFastBins>(scores, tiles[ti], edge_sub, W, out, todo, left, lds_fast);
        __syncthreads();
    }
}

__global__ __launch_bounds__(kAdjThreads) void adjust_mean_kernel(const double* __restrict__ scores,
                                                                   const AdjustTile* __restrict__ tiles,
                                                                   const double* __restrict__ edge_sub, int W,
                                                                   double* __restrict__ out) {
    extern __shared__ unsigned long long lds_keys[];
    double* x = (double*)lds_keys;
    const AdjustTile t = tiles[blockIdx.x];
    const double sub = edge_sub ? edge_sub[t.interval] : 0.0;
    const double* in = scores + t.in_base;
    const int n_in = t.n_out + W - 1;
    for (int i = threadIdx.x; i < n_in; i += kAdjThreads) x[i] = in[i] - sub;
    __syncthreads();
    for (int o = threadIdx.x; o < t.n_out; o += kAdjThreads) {
        double s = 0.0;
        for (int j = 0; j < W; ++j) s += x[o + j];
        out[t.out_base + o] = x[o + W / 2] - s / (double)W;
    }
}

// One thread per output.  Interior points follow scipy.ndimage's symmetric
// correlate1d order (centre tap first, then outermost pair inwards) with no
// fused multiply-add, so they reproduce savgol_filter bit for bit; the first
// and last `half` points apply the polynomial edge fit as a half x window matrix.
__global__ __launch_bounds__(kAdjThreads) void savgol_kernel(const double* __restrict__ adj,
                                                              const AdjustTile* __restrict__ tiles,
                                                              const double* __restrict__ coef,
                                                              const double* __restrict__ edge, int sw,
                                                              double* __restrict__ out) {
    const AdjustTile t = tiles[blockIdx.x];
    const int half = sw / 2;
    const double* x = adj + (t.out_base - t.o0);  // interval's first adjusted value
    for (int i = threadIdx.x; i < t.n_out; i += kAdjThreads) {
        const int o = t.o0 + i;
        double acc;
        if (o < half) {
            const double* e = edge + (size_t)o * sw;
            acc = 0.0;
            for (int j = 0; j < sw; ++j) acc = __dadd_rn(acc, __dmul_rn(e[j], x[j]));
        } else if (o >= t.m - half) {
            const double* e = edge + (size_t)(half + o - (t.m - half)) * sw;
            const double* xr = x + (t.m - sw);
            acc = 0.0;
            for (int j = 0; j < sw; ++j) acc = __dadd_rn(acc, __dmul_rn(e[j], xr[j]));
        } else {
            acc = __dmul_rn(x[o], coef[half]);
            for (int jj = -half; jj < 0; ++jj)
                acc = __dadd_rn(acc, __dmul_rn(__dadd_rn(x[o + jj], x[o - jj]), coef[half + jj]));
        }
        out[t.out_base + i] = acc;
    }
}

}  // namespace

__device__ __forceinline__ int wave_incl_scan_dpp(int x) {  // inclusive prefix sum over the 64 lanes (row shifts + row broadcasts)
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}

// ---- the tile lists, built where they are used --------------------------------------------------------------------
// (one 32-byte descriptor per tile: 40 000 + 10 000 of them for 10 000 runs of 5 kb were 1.6 MB of pageable uploads
// and ~70 us of the stream before the first kernel of a 0.5 ms call; the run offsets are 80 KB)
// pre[k][i]: tiles of kind k (0: sort kernel, 1: histogram median) before run i; one block, runs 1024 at a time.
// offs: the run offsets where the caller put them - page-locked HOST memory the device reads across the link (80 KB for
// 10 000 runs: a copy of its own in front of this kernel was 7 us of DMA, 9 us of gap and a dozen of engine-to-engine
// synchronisation); offs_dev: the device copy the other kernels read, written here.
__global__ __launch_bounds__(1024) void adjust_tile_counts_kernel(const int64_t* __restrict__ offs, int64_t* __restrict__ offs_dev,
                                                                   int n_iv, int W, int tile0, int tile1, int* __restrict__ pre0,
                                                                   int* __restrict__ pre1, int* __restrict__ todo) {
    // kR trips of 1024 runs are LOADED together (coalesced: thread t takes run r * 1024 + t of every trip) and then
    // scanned one after the other: a trip per load was a memory latency per 1024 runs (23 us for 10 000), a thread's
    // kR consecutive runs were 64 cache lines per load instruction on the one CU this block has (20 us).  Also clears
    // the marks of the histogram kernels.
    constexpr int kR = 16;
    // ceil(m / t) for m < 2^31 without an integer division: the double product is within 1e-7 of the quotient, so its
    // floor is right or one short; the remainder settles it
    const double rcp0 = 1.0 / (double)tile0, rcp1 = tile1 ? 1.0 / (double)tile1 : 0.0;
    auto ceil_div = [](unsigned int m, unsigned int t, double rcp) -> unsigned int {
        const unsigned int x = m + t - 1u;
        unsigned int q = (unsigned int)((double)x * rcp);
        if (x - q * t >= t) ++q;
        return q;
    };
    // One pass of kR trips: every wave scans its 64 runs of each trip, the kR x 16 wave totals are scanned ONCE by the
    // first four waves, and every thread adds its wave's base - four barriers per 16 384 runs (a barrier trio and a
    // serial walk over the wave totals per trip were 19 of this kernel's 25 us for 10 000 runs).
    __shared__ int wave_tot[2][kR * 16];  // [kind][trip * 16 + wave]: the wave's total, then its exclusive base
    __shared__ int quad_tot[2][4];
    __shared__ int carry[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 2) carry[tid] = 0;
    if (todo)
        for (int i = tid; i < n_iv + 2; i += 1024) todo[i] = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n_iv; i0 += 1024 * kR) {
        int64_t o[kR], o1[kR];
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int i = min(i0 + r * 1024 + tid, n_iv - 1);
            o[r] = offs[i];
            o1[r] = offs[i + 1];
        }
        if (offs_dev) {
#pragma unroll
            for (int r = 0; r < kR; ++r) {
                const int i = i0 + r * 1024 + tid;
                if (i < n_iv) offs_dev[i] = o[r];
                if (i == n_iv - 1) offs_dev[n_iv] = o1[r];
            }
        }
        int c0[kR], c1[kR], x0[kR], x1[kR];
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int i = i0 + r * 1024 + tid;
            c0[r] = c1[r] = 0;
            if (i < n_iv) {
                const unsigned int m = (unsigned int)(o1[r] - o[r] - W);  // (0 .. INT32_MAX, checked by the caller)
                c0[r] = (int)ceil_div(m, (unsigned)tile0, rcp0);
                c1[r] = tile1 ? (int)ceil_div(m, (unsigned)tile1, rcp1) : 0;
            }
            x0[r] = wave_incl_scan_dpp(c0[r]);  // (`__shfl_up` is an LDS permute: twelve dependent ones per trip were most of this kernel)
            x1[r] = wave_incl_scan_dpp(c1[r]);
            if (lane == 63) { wave_tot[0][r * 16 + wv] = x0[r]; wave_tot[1][r * 16 + wv] = x1[r]; }
        }
        __syncthreads();
        if (tid < kR * 16) {  // the 256 wave totals, in run order: exclusive scan by four waves
            const int t0 = wave_tot[0][tid], t1 = wave_tot[1][tid];
            const int s0 = wave_incl_scan_dpp(t0), s1 = wave_incl_scan_dpp(t1);
            if (lane == 63) { quad_tot[0][wv] = s0; quad_tot[1][wv] = s1; }
            wave_tot[0][tid] = s0 - t0;  // (exclusive within the quarter; the quarters in front are added below)
            wave_tot[1][tid] = s1 - t1;
        }
        __syncthreads();
        if (tid < kR * 16) {
            int b0 = 0, b1 = 0;
            for (int j = 0; j < wv; ++j) { b0 += quad_tot[0][j]; b1 += quad_tot[1][j]; }
            wave_tot[0][tid] += b0;
            wave_tot[1][tid] += b1;
        }
        __syncthreads();
        const int base0 = carry[0], base1 = carry[1];
#pragma unroll
        for (int r = 0; r < kR; ++r) {
            const int i = i0 + r * 1024 + tid;
            if (i < n_iv) {
                pre0[i] = base0 + wave_tot[0][r * 16 + wv] + x0[r] - c0[r];
                pre1[i] = base1 + wave_tot[1][r * 16 + wv] + x1[r] - c1[r];
            }
        }
        __syncthreads();
        if (tid == 0) {
            carry[0] = base0 + quad_tot[0][0] + quad_tot[0][1] + quad_tot[0][2] + quad_tot[0][3];
            carry[1] = base1 + quad_tot[1][0] + quad_tot[1][1] + quad_tot[1][2] + quad_tot[1][3];
        }
        __syncthreads();
    }
    if (tid == 0) { pre0[n_iv] = carry[0]; pre1[n_iv] = carry[1]; }
}

// tile j of kind k: its run by bisection of pre[k], then the descriptor (ftk_wps_adjust's own arithmetic)
__global__ __launch_bounds__(256) void adjust_tile_fill_kernel(const int64_t* __restrict__ offs, int n_iv, int W, int tile0,
                                                                int tile1, const int* __restrict__ pre0,
                                                                const int* __restrict__ pre1, AdjustTile* __restrict__ t0,
                                                                AdjustTile* __restrict__ t1) {
    const int j = blockIdx.x * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int* pre = k ? pre1 : pre0;
        const int tile = k ? tile1 : tile0;
        AdjustTile* dst = k ? t1 : t0;
        if (!tile || !dst || j >= pre[n_iv]) continue;
        int lo = 0, hi = n_iv;  // largest i with pre[i] <= j
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (pre[mid] <= j) lo = mid; else hi = mid;
        }
        const int64_t o0 = (int64_t)(j - pre[lo]) * tile, m = offs[lo + 1] - offs[lo] - W;
        AdjustTile t;
        t.in_base = offs[lo] + o0;
        t.out_base = offs[lo] - (int64_t)lo * W + o0;
        t.n_out = (int32_t)min((int64_t)tile, m - o0);
        t.o0 = (int32_t)o0;
        t.m = (int32_t)m;
        t.interval = lo;
        dst[j] = t;
    }
}

int adjust_sort_size(int W, int* tile_out) {
    int n = 1;
    while (n < 2 * W) n <<= 1;
    n = n < 256 ? 256 : n;
    if (tile_out) *tile_out = n - W + 1;
    return n;
}

void launch_adjust_tile_counts(hipStream_t s, const int64_t* offs_src, int64_t* offs, int n_iv, int W, int tile0, int tile1,
                               int* pre0, int* pre1, int* todo) {
    adjust_tile_counts_kernel<<<1, 1024, 0, s>>>(offs_src, offs, n_iv, W, tile0, tile1, pre0, pre1, todo);
}

void launch_adjust_tile_fill(hipStream_t s, const int64_t* offs, int n_iv, int W, int tile0, int tile1, const int* pre0,
                             const int* pre1, AdjustTile* t0, int n0, AdjustTile* t1, int n1) {
    const int n = n0 > n1 ? n0 : n1;
    if (n > 0) adjust_tile_fill_kernel<<<(n + 255) / 256, 256, 0, s>>>(offs, n_iv, W, tile0, t1 ? tile1 : 0, pre0, pre1, t0, t1);
}

void launch_adjust_filter(hipStream_t s, const double* scores, const AdjustTile* tiles, int n_tiles,
                          const double* edge_sub, int W, int use_mean, double* out, const AdjustTile* fast_tiles,
                          int n_fast_tiles, int* todo, int n_iv) {
    int tile;
    const int n_sort = adjust_sort_size(W, &tile);
    if (use_mean) {
        const size_t lds = (size_t)(tile + W - 1) * 8;
        adjust_mean_kernel<<<n_tiles, kAdjThreads, lds, s>>>(scores, tiles, edge_sub, W, out);
        return;
    }
    const bool hist = fast_tiles && todo;
    int* left = hist ? todo + n_iv : nullptr;  // [2]: something marked kTodoWide / kTodoSort (cleared with todo by launch_adjust_tiles)
    if (hist) {  // integers in a narrow range: sliding histograms; what they cannot take is marked ...
        const size_t bytes_sv = (size_t)(kAdjustFastTile + W - 1 + 3) / 4 * 4 + 8;  // (+8: a run's last trip reads whole dwords)
        adjust_median_hist_kernel<128><<<n_fast_tiles, kFastThreads, (size_t)128 / 2 * 64 * 4 + bytes_sv, s>>>(scores, fast_tiles, n_fast_tiles, edge_sub, W, out, todo, left);
        adjust_median_hist_kernel<256><<<std::min(n_fast_tiles, 1024), kFastThreads, (size_t)256 / 2 * 64 * 4 + bytes_sv, s>>>(scores, fast_tiles, n_fast_tiles, edge_sub, W, out, todo, left);
    }
    // ... and sorted (every interval when there is no histogram pass)
    const size_t lds = (size_t)n_sort * (8 + 4);
    adjust_median_kernel<<<n_tiles, kAdjThreads, lds, s>>>(scores, tiles, n_tiles, edge_sub, W, n_sort, out, hist ? todo : nullptr, left);
}

void launch_savgol(hipStream_t s, const double* adj, const AdjustTile* tiles, int n_tiles, const double* coef,
                   const double* edge, int sw, double* out) {
    savgol_kernel<<<n_tiles, kAdjThreads, 0, s>>>(adj, tiles, coef, edge, sw, out);
}

}  // namespace ftk
