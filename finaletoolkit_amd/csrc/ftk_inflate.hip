// DEFLATE (RFC 1951) decoder for BGZF blocks on gfx950: one wavefront per block (see ftk_inflate.h).
//
// Shape of the work.  A BGZF block is an independent DEFLATE stream of at most 64 KB of data; a 48 MB piece of a
// fragment file holds ~3 000 of them.  Inside a block the Huffman decoding is a serial chain (the position of a
// code is only known once the previous one has been decoded), so it runs as UNIFORM work of the wave: the bit
// buffer, positions and table entries live in scalar registers (values fetched from LDS / VGPRs come back through
// v_readfirstlane / v_readlane), and the 64 lanes are used where there is width: building the decoding tables
// (ballot-ranked canonical codes, one symbol per lane), the LZ77 copies (up to 64 bytes per step), the input
// prefetch (256 bytes of the payload per vector load, handed to the bit buffer word by word with v_readlane) and
// the write-behind of finished output.  The chip is filled by running thousands of such waves side by side:
// 7.5 KB of LDS per wave (kRing = 2048, 9-bit literal root) -> 20 waves per CU.
// Since round 3 the symbols themselves are decoded by the lanes too, a window of 64 bit offsets at a time (see the
// symbol loop): the scalar chain only walks the symbols' bit lengths.
//
// Output window.  The last kRing bytes (2 KB by default) of a block's output live in an LDS ring indexed by the
// absolute output address, so nearly every match (fragment rows repeat the previous line, 30-60 bytes back) is an
// LDS-to-LDS copy; finished kGran-byte granules (1 KB: half the ring) are streamed to HBM with 16-byte stores as
// soon as the write position passes them, and a match that reaches further back than kFarDist (1 726 bytes) reads
// the bytes it needs from HBM.  Why that is safe (the static_asserts below pin it): such a match's newest source
// byte is A0 + len - 1 - d <= A0 - (kFarDist - 256), and every granule below floor(A0 / kGran) * kGran >=
// A0 - (kGran - 1) has been flushed, so with kFarDist - 257 >= kGran - 1 all of its sources are in HBM (behind a
// workgroup fence); and an LDS-to-LDS match (d <= kFarDist) never has its sources overwritten by its own stores,
// which reach at most 258 + 64 bytes past A0: kFarDist + 258 + 64 <= kRing.
#include <hip/hip_runtime.h>
#include <type_traits>

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>

#include "ftk_inflate.h"

namespace ftk {
namespace {

// The output window held in LDS.  2 KB: 9 KB of LDS per wave, 17 waves per CU - on pieces that fill the chip (BAM, text
// streams with two pieces in flight) 17 % / 8 % faster than 4 KB (14 waves per CU): matches further back than ~1.7 KB
// read HBM instead, rare in fragment rows and BAM records.  (A launch of fewer blocks than the chip holds is 2 % slower.)
#ifndef FTK_INFLATE_RING
#define FTK_INFLATE_RING 2048
#endif
// Root of the literal / length look-up: 9 bits = a 2 KB pair table, 7.5 KB of LDS per wave, 20 waves per CU (then the
// 87 VGPRs bound it).  With the windowed symbol loop the waves a CU holds set the throughput of a chip-filling launch:
// 10 bits (4 KB, 17 waves) 9.5 ms per 650 MB of text and 12.7 ms per 590 MB of BAM records, 9 bits 7.6 / 10.4 ms;
// a launch of fewer blocks than the chip holds is unchanged (3.0 ms: one block's chain).  Forcing 24 waves (78 VGPRs,
// distance root 8) gains nothing more, 25 waves at 72 VGPRs lose 5 % (tools/inflate_variants.sh).  Codes longer than
// the root - rare symbols - end a window and take the serial path.
#ifndef FTK_INFLATE_ROOT
#define FTK_INFLATE_ROOT 9
#endif
constexpr int kRing = FTK_INFLATE_RING, kRingMask = kRing - 1;
constexpr int kGranShift = kRing >= 8192 ? 11 : 10, kGran = 1 << kGranShift;  // write-behind granule: at most half the ring
#ifndef FTK_INFLATE_WINDOWED
#define FTK_INFLATE_WINDOWED 1
#endif
// Root of the distance look-up: 8 bits (1 KB; 9 until round 5).  Distance codes longer than the root are walked bit by bit
// (long_code / decode_long) - rare - and the smaller table is worth 10 % on chip-filling launches of the lane-parallel
// loop (fragment rows 5.84 -> 5.25 ms; 7 bits the same again).
#ifndef FTK_INFLATE_DIST_ROOT
#define FTK_INFLATE_DIST_ROOT 8
#endif
// A window's literal and match bytes resolved by the lanes side by side (see the symbol loop): a property of the LAUNCH
// (template parameter VEC of the kernel; inflate_launch's vector_matches).  Measured on chip-filling launches
// (tools/inflate_variants.sh, round 4): BAM records 10.17 -> 9.70 ms per 590 MB (-5 %: mostly literals, the few matches -
// read names, flags - far apart), fragment rows 7.52 -> 7.62 ms per 650 MB and 3.39 -> 3.59 ms on a launch of fewer blocks
// than the chip holds (+1 % / +6 %: three short matches per window whose serial copies cost no more than the
// byte-to-symbol expansion) - so BAM streams ask for it and text streams do not.  FTK_INFLATE_VECMATCH=0/1 forces it
// off / on for every launch (A/B builds).
constexpr int kLitRoot = FTK_INFLATE_ROOT, kDistRoot = FTK_INFLATE_DIST_ROOT, kPreRoot = 7;
constexpr int kFarDist = kRing - 258 - 64;    // matches further back than this read from HBM
static_assert((kRing & (kRing - 1)) == 0 && kGran <= kRing / 2, "the write-behind granule must be at most half the ring");
static_assert(kFarDist + 258 + 64 <= kRing, "an LDS-to-LDS match must not overwrite its own sources");
static_assert(kFarDist - 257 >= kGran - 1, "the sources of a far match must already be flushed to HBM");

// Bytes one window of the symbol loop may produce: its literals are stored before its matches copy, up to kWinCap - 1
// bytes ahead of the write position, and must not land on ring slots an LDS-to-LDS match of the window still reads
// (>= kFarDist back from ITS start, which is not before the window's): kFarDist + kWinCap < kRing.
constexpr int kWinCap = kRing - kFarDist - 2;
static_assert(kFarDist + kWinCap < kRing && kWinCap >= 258, "a window's literals must not overwrite a match's sources");
static_assert(kFarDist - 256 - kWinCap >= kGran - 1, "the sources of a window's far matches must already be flushed");

__device__ __forceinline__ int wave_incl_scan(int x) {  // inclusive prefix sum over the 64 lanes (DPP)
    x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
    x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
    return x;
}

struct __align__(16) WaveLds {
    uint8_t ring[kRing];
    uint32_t pair[1 << kLitRoot];    // the look-up of the symbol loop: up to TWO literals per entry (see build_pairs);
                                     // its first half doubles as the one-symbol table (uint16: symbol << 4 | code
                                     // length, 0 = longer than the root) while a block's tables are being built
    uint32_t dist[1 << kDistRoot];   // code length | extra bits << 4 | base distance << 8 (the first half holds the
                                     // one-symbol uint16 table while it is being built); 0 = longer than the root
    uint16_t pre[1 << kPreRoot];
    uint16_t sorted[288 + 32];       // symbols in canonical order (by length, then value): lit/len at 0, distance at 288
    uint16_t cnt[2][16];             // symbols per code length
    uint16_t next_code[16], offs[16];
    uint8_t lens[320 + 19];
};

#define UNI(x) __builtin_amdgcn_readfirstlane((int)(x))

struct Bits {
    const uint32_t* w;  // the compressed buffer as aligned words
    uint32_t end_word;  // first word index behind the payload
    uint32_t widx;      // next word to append
    uint32_t cbase;     // word index held by lane 0 of `cache`
    uint32_t cache;     // per lane: w[cbase + lane]
    uint64_t buf;
    int cnt;

    __device__ __forceinline__ void load_cache(int lane) {
        const uint32_t i = cbase + (uint32_t)lane;
        cache = i < end_word ? w[i] : 0u;
    }
    __device__ __forceinline__ void refill(int lane) {  // afterwards at least 32 valid bits
        if (cnt < 32) {
            if (widx - cbase >= 64u) {
                cbase = widx;
                load_cache(lane);
            }
            const uint32_t v = (uint32_t)__builtin_amdgcn_readlane((int)cache, (int)(widx - cbase));
            ++widx;
            buf |= (uint64_t)v << cnt;
            cnt += 32;
        }
    }
    // absolute position (in bits from the start of the compressed buffer) of the next unread bit
    __device__ __forceinline__ uint32_t bitpos() const { return widx * 32u - (uint32_t)cnt; }
    __device__ __forceinline__ void seek(uint32_t bp, int lane) {  // afterwards 33 .. 64 valid bits from bit bp on
        const uint32_t w0 = bp >> 5;
        if (w0 - cbase > 62u) {  // (also w0 < cbase: a refill moved the cache past bits still in the buffer)
            cbase = w0;
            load_cache(lane);
        }
        const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)cache, (int)(w0 - cbase));
        const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)cache, (int)(w0 + 1u - cbase));
        const int sh = (int)(bp & 31u);
        buf = (((uint64_t)hi << 32) | lo) >> sh;
        cnt = 64 - sh;
        widx = w0 + 2u;
    }
    __device__ __forceinline__ uint32_t peek(int n) const { return (uint32_t)buf & ((1u << n) - 1u); }
    __device__ __forceinline__ void drop(int n) { buf >>= n; cnt -= n; }
    __device__ __forceinline__ uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
};

// Canonical Huffman decoding tables from code lengths lens[0, n) (in LDS): `table` (1 << root entries) resolves
// codes up to `root` bits in one look-up; `sorted` / `cnt` serve the bit-by-bit path for longer ones.  Returns
// false for an over-subscribed set of lengths.  Lane-parallel: one symbol per lane, ranks among equal lengths from
// ballots.
__device__ __forceinline__ bool build_table(WaveLds& L, const uint8_t* lens, int n, int root, uint16_t* table, int which, int lane) {
    const int size = 1 << root;
    for (int i = lane; i < size; i += 64) table[i] = 0;
    int count[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) count[l] = 0;
    for (int base = 0; base < n; base += 64) {
        const int sym = base + lane;
        const int l = sym < n ? lens[sym] : 0;
#pragma unroll
        for (int k = 1; k < 16; ++k) count[k] += __popcll(__ballot(l == k));
    }
    int left = 1, code = 0, off = 0;
    bool ok = true;
#pragma unroll
    for (int k = 1; k < 16; ++k) {
        left = (left << 1) - count[k];
        ok = ok && left >= 0;
        code = (code + count[k - 1]) << 1;  // count[0] is 0 here
        if (lane == 0) {
            L.next_code[k] = (uint16_t)code;
            L.offs[k] = (uint16_t)off;
            L.cnt[which][k] = (uint16_t)count[k];
        }
        off += count[k];
    }
    if (!ok) return false;
    int run[16];
#pragma unroll
    for (int l = 0; l < 16; ++l) run[l] = 0;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = 0; base < n; base += 64) {
        const int sym = base + lane;
        const int l = sym < n ? lens[sym] : 0;
        int rank = 0;
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            const unsigned long long m = __ballot(l == k);
            if (l == k) rank = run[k] + __popcll(m & lt);
            run[k] += __popcll(m);
        }
        if (l) {
            const unsigned c = (unsigned)L.next_code[l] + (unsigned)rank;
            L.sorted[which * 288 + L.offs[l] + rank] = (uint16_t)sym;
            if (l <= root) {
                const unsigned rev = __brev(c) >> (32 - l);  // codes are packed most significant bit first
                const uint16_t e = (uint16_t)((sym << 4) | l);
                for (unsigned k = rev; k < (unsigned)size; k += 1u << l) table[k] = e;
            }
        }
    }
    return true;
}

// The symbol loop's table.  The decode is bound by instruction issue (a wave issues one instruction every four
// cycles, and the chain per symbol is look-up, classify, consume, store), so the table resolves as much as 11
// bits can hold in ONE look-up: entry = total bits | kind << 5 | a << 8 | b << 20 with
//   kind 2: two literals a, b (both codes fit in the 11 bits: the common case for text, whose frequent
//           characters have 4-6 bit codes);  kind 1: one literal a;
//   kind 0: a = a length symbol or end-of-block (256..285); entry 0: a code longer than 11 bits (or none).
__device__ __forceinline__ void build_pairs(WaveLds& L, int lane) {
    constexpr int size = 1 << kLitRoot;
    const uint16_t* one = reinterpret_cast<const uint16_t*>(L.pair);  // the one-symbol table, converted in place:
    // entry e of the pair table covers one-symbol entries 2e and 2e+1, and needs entries e and e >> l1 <= e / 2,
    // so going down from the top no step overwrites what a later one reads (within a step every lane has read
    // before any lane writes)
    for (int base = size - 64; base >= 0; base -= 64) {
        const int e = base + lane;
        const unsigned t = one[e];
        const unsigned l1 = t & 15u, s1 = t >> 4;
        unsigned v = 0;
        if (l1) {
            if (s1 >= 256u) {  // end of block (base 0) or a length symbol: base length and extra bits in the entry
                const unsigned ls = s1 - 257u;
                unsigned base = 0, eb = 0;
                if (s1 == 256u) base = 0;
                else if (ls < 8u) base = ls + 3u;
                else if (ls == 28u) base = 258u;
                else if (ls < 28u) { eb = (ls >> 2) - 1u; base = ((4u + (ls & 3u)) << eb) + 3u; }
                else base = 511u;  // 286 / 287: no such length (the decoder reports it)
                v = l1 | (base << 8) | (eb << 20);
            } else {
                v = l1 | (1u << 5) | (s1 << 8);
                const unsigned t2 = one[e >> l1];  // the bits behind the first code, zero-extended: valid for a code
                const unsigned l2 = t2 & 15u, s2 = t2 >> 4;  // that fits into what is left of the root bits
                if (l2 && l1 + l2 <= (unsigned)kLitRoot && s2 < 256u) v = (l1 + l2) | (2u << 5) | (s1 << 8) | (s2 << 20);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        L.pair[e] = v;
    }
}

// the distance table in place: uint16 symbol << 4 | length -> length | extra bits << 4 | base << 8
__device__ __forceinline__ unsigned dist_entry(unsigned l, unsigned ds) {
    unsigned base, eb = 0;
    if (ds < 4u) base = ds + 1u;
    else if (ds < 30u) { eb = (ds >> 1) - 1u; base = ((2u + (ds & 1u)) << eb) + 1u; }
    else base = 0x7fffffu;  // 30 / 31: no such distance
    return l | (eb << 4) | (base << 8);
}
__device__ __forceinline__ void build_dist(WaveLds& L, int lane) {
    constexpr int size = 1 << kDistRoot;
    const uint16_t* one = reinterpret_cast<const uint16_t*>(L.dist);
    for (int base = size - 64; base >= 0; base -= 64) {
        const int e = base + lane;
        const unsigned t = one[e];
        const unsigned l = t & 15u;
        const unsigned v = l ? dist_entry(l, t >> 4) : 0u;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        L.dist[e] = v;
    }
}

// one code from the bit buffer (>= 15 valid bits), length by length (codes longer than a table's root)
__device__ __forceinline__ int decode_long(const WaveLds& L, Bits& b, int which) {
    unsigned v = (unsigned)b.buf;
    int code = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; ++len) {
        code |= (int)(v & 1u);
        v >>= 1;
        const int count = UNI(L.cnt[which][len]);
        if (code - count < first) {
            b.drop(len);
            return UNI(L.sorted[which * 288 + index + (code - first)]);
        }
        index += count;
        first = (first + count) << 1;
        code <<= 1;
    }
    return -1;
}

#ifdef FTK_INFLATE_PROFILE
// tools/inflate_symbol_cost.py (library built with -DFTK_INFLATE_TIMING -DFTK_INFLATE_PROFILE): shader-clock cycles the
// waves spent in each section of the symbol loop, and how often they went through it
__device__ unsigned long long g_prof[32];
#define PROF_DECL unsigned long long prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_n[8] = {0, 0, 0, 0, 0, 0, 0, 0}, prof_last = 0
#define PROF(i, dep)                                                                                                   \
    {                                                                                                                  \
        unsigned long long t_;                                                                                         \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), "+v"(dep)::"memory"); \
        prof_acc[i] += t_ - prof_last;                                                                                 \
        prof_n[i] += 1;                                                                                                \
        prof_last = t_;                                                                                                \
    }
#else
#define PROF_DECL
#define PROF(i, dep)
#endif

#ifdef FTK_INFLATE_TIMING
// tools/inflate_block_times.py builds the library with this: start / end of every block's wavefront (100 MHz ticks)
__device__ unsigned long long g_block_ticks[2 * 65536];
#endif

#ifdef FTK_INFLATE_WAVES
#define FTK_INFLATE_OCC __attribute__((amdgpu_waves_per_eu(FTK_INFLATE_WAVES, FTK_INFLATE_WAVES)))
#else
#define FTK_INFLATE_OCC
#endif
// ---- lane-parallel symbol decoding (round 5; template flag LANES, inflate_launch's lane_scratch) ----------
// The windows above spend their vector work on 64 bit offsets of which ~9 start a symbol, and the chain through those
// starts is serial.  Here the lanes take DIFFERENT stretches of the input instead: a super-window is 64 stretches of
// kLaneBits bits; lane l decodes the symbols of its stretch one after the other, all lanes in lock step - 64 symbols per
// trip through the loop.  Only lane 0 knows where its first symbol starts; the others start at their stretch's first bit,
// which is most likely the middle of a symbol, and rely on what prefix codes do: a decoder that starts anywhere falls
// into step with the true symbol sequence after a few symbols.  So: (A) every lane decodes its stretch from its first
// bit, noting the positions it took for symbol starts (a checkpoint per 32-bit word, LDS) and writing its symbols as
// 32-bit tokens (literals or length + distance) to a scratch region; (B) lane l's TRUE first symbol starts where lane
// l - 1's last one ends, so the lanes pass their end positions to the right and every lane decodes again from its true
// start until it lands on a position it has visited - from there on its tokens of (A) are the right ones - repeated
// while any end position still moves (a lane that never falls into step inside its stretch moves its end and so its right
// neighbour's start); (C) every lane keeps where its valid tokens are, fetch() finds the owner of the t-th token; (D) 64
// tokens at a time, a prefix sum gives every token its place in the output and the group's BYTES are resolved side by
// side (keys at symbol starts, running maximum, pointer doubling - see there).
// A lane stops at anything the tables do not resolve in one look-up (end of block, a bit pattern that is no code): the
// super-window ends in front of it and ONE window of the older kind takes it.
#ifndef FTK_LANE_BITS
#define FTK_LANE_BITS 768   // (512 / 640 / 768 / 896 / 1024 measured: fragment rows the same to 1 % up to 768 and 4-5 % slower
#define FTK_LANE_TOK 256    // beyond, BAM records 5.27 / 5.06 / 5.03 / 5.15 / 5.19 ms per chip-filling launch)
#define FTK_LANE_CATCH 96
#endif
// Bytes a group of (D) may produce.  With the group's bytes resolved side by side nothing is stored before every source has
// been read, so a window's kWinCap does not bind; what does: a far match's sources must have been written behind (byte j of
// the group reads A + j - d < A + j - kFarDist, and everything in front of A - (kGran - 1) is in HBM: j < kFarDist - kGran
// + 2), and the write-behind takes one granule per group (T <= kGran).  256 .. 640 measured: 384 is the flat minimum.
#ifndef FTK_LANE_CAP
#define FTK_LANE_CAP 384
#endif
constexpr int kLaneCap = FTK_LANE_CAP;
static_assert(kLaneCap % 64 == 0 && kLaneCap <= kFarDist - kGran + 2 && kLaneCap <= kGran, "see above");
constexpr int kLaneBits = FTK_LANE_BITS;          // bits of input per lane and super-window
constexpr int kLaneTok = FTK_LANE_TOK;            // tokens a lane may write in (A); more ends its stretch early
constexpr int kLaneCatch = FTK_LANE_CATCH;        // ... and on its way into step in (B)
constexpr int kLaneRounds = 6;                    // passes of (B) before the super-window is cut at the first unsettled lane
constexpr int kLaneSlots = 4096;                  // scratch slots: as many as the chip holds of these waves (256 CUs x <= 16)
constexpr size_t kLaneSlotWords = 64 * kLaneTok + 64 * kLaneCatch;  // a lane's tokens of (A), and of its way into step
// LDS of the loop beside the tables: the symbol starts of (A) - a 16-bit checkpoint per 32-bit word of a lane's stretch -
// and, in (C) / (D), the owner slots (words 0..63) and a group's rows of byte states (from word 192)
constexpr int kLaneStartWords = 64 * (kLaneBits / 32) / 2;
constexpr int kLaneLdsWords = kLaneStartWords > 192 + (kLaneCap / 64) * 64 ? kLaneStartWords : 192 + (kLaneCap / 64) * 64;
static_assert(kLaneTok < 1024, "a checkpoint holds a token number in ten bits");
struct LaneScratch {
    unsigned busy[kLaneSlots];
    uint32_t words[1];  // kLaneSlots x kLaneSlotWords
};

#ifdef FTK_LANES_STATS
// tools/lanes_stats.py (library built with -DFTK_LANES_STATS): what the super-windows of a launch looked like
__device__ unsigned long long g_lanes_stats[32];
#define LSTAT(i, v) do { if (lane == 0) atomicAdd(&g_lanes_stats[i], (unsigned long long)(v)); } while (0)
// (times are kept in registers and added once a super-window: an atomic per mark costs more than what it marks)
#if FTK_LANES_STATS > 1
#define LT_N 14
#else
#define LT_N 4
#endif
#define LTIME_DECL unsigned lt_last = (unsigned)wall_clock64(), lt_acc[LT_N] = {}; unsigned l_trips_a = 0, l_trips_b = 0, l_rounds = 0, l_need2 = 0, l_nojoin1 = 0, l_trips_b1 = 0
#define LTIME(i) do { const unsigned t_ = (unsigned)wall_clock64(); lt_acc[(i) - 10] += t_ - lt_last; lt_last = t_; } while (0)
#define BT_DECL unsigned bt_last = (unsigned)wall_clock64(), bt_acc[3] = {0, 0, 0}
#define BTIME(i) do { const unsigned t_ = (unsigned)wall_clock64(); bt_acc[i] += t_ - bt_last; bt_last = t_; } while (0)
#define BT_FLUSH do { LSTAT(30, bt_acc[0]); LSTAT(31, bt_acc[1]); LSTAT(14, bt_acc[2]); LSTAT(15, 1); } while (0)
#define LACC_DECL unsigned d_matches = 0, d_groups = 0, d_rounds = 0
#define LACC(x, v) x += (unsigned)(v)
#if FTK_LANES_STATS > 1
#define LTIME_FLUSH_D do { LSTAT(16, lt_acc[6]); LSTAT(17, lt_acc[7]); LSTAT(18, lt_acc[8]); LSTAT(19, lt_acc[9]); LSTAT(20, lt_acc[10]); LSTAT(27, lt_acc[11]); LSTAT(28, lt_acc[12]); LSTAT(29, lt_acc[13]); } while (0)
#else
#define LTIME_FLUSH_D
#endif
#define LTIME_FLUSH do { LSTAT(10, lt_acc[0]); LSTAT(11, lt_acc[1]); LSTAT(12, lt_acc[2]); LSTAT(13, lt_acc[3]); LSTAT(5, l_trips_a); LSTAT(6, l_trips_b); LSTAT(3, l_rounds); LSTAT(22, l_need2); LSTAT(23, l_nojoin1); LSTAT(24, l_trips_b1); LTIME_FLUSH_D; } while (0)
#if FTK_LANES_STATS > 1   // (markers inside (D): they cost more than what they time, so only on request)
#define LTIME_D(i) LTIME(i)
#else
#define LTIME_D(i)
#endif
#else
#define LTIME_D(i)
#define LSTAT(i, v)
#define LTIME_DECL
#define LTIME_FLUSH
#define LACC_DECL
#define LACC(x, v)
#define BT_DECL
#define BTIME(i)
#define BT_FLUSH
#define LTIME(i)
#endif

template <bool VEC, bool LANES>
__global__ __launch_bounds__(64) FTK_INFLATE_OCC void bgzf_inflate_kernel(const uint8_t* __restrict__ comp,
                                                          const InflateBlock* __restrict__ tab, int n_blocks,
                                                          uint8_t* __restrict__ out, InflateStatus* __restrict__ status,
                                                          LaneScratch* __restrict__ lane_scratch) {
    __shared__ WaveLds L;
    __shared__ uint32_t lanes_vis[LANES ? kLaneLdsWords : 1];
    const int lane = threadIdx.x;
    const int blk = blockIdx.x;
    if (blk >= n_blocks) return;
    uint32_t* lane_tok = nullptr;  // this wave's scratch slot (LANES)
    int lane_slot = -1;
    if (LANES) {
        // a slot nobody holds: there are as many as the chip can hold waves, so one is free; probing starts at a place
        // of the block's own
        // (a wave that finds none in two rounds over all of them - which the slot count rules out - does not spin on:
        // it decodes its block with windows of the older kind alone)
        unsigned s0 = ((unsigned)blk * 2654435761u) >> 20;  // 12 bits
        int got = 0;
        if (lane == 0) {
            for (int tries = 0; tries < 2 * kLaneSlots; ++tries, s0 = (s0 + 1u) & (unsigned)(kLaneSlots - 1))
                if (atomicCAS(&lane_scratch->busy[s0], 0u, 1u) == 0u) {
                    got = 1;
                    break;
                }
        }
#ifdef FTK_LANES_NO_SLOT  // (test builds: every wave takes the path of a wave that found no slot)
        if (got && lane == 0) atomicExch(&lane_scratch->busy[s0], 0u);
        got = 0;
#endif
        if (UNI(got)) {
            lane_slot = UNI(s0);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            lane_tok = lane_scratch->words + (size_t)lane_slot * kLaneSlotWords;
        }
    }
#ifdef FTK_INFLATE_TIMING
    if (lane == 0 && blk < 65536) g_block_ticks[2 * blk] = wall_clock64();
#endif
    const uint32_t in_off = tab[blk].in_off, in_len = tab[blk].in_len;
    const uint32_t out_off = tab[blk].out_off, out_len = tab[blk].out_len;
    Bits b;
    b.w = reinterpret_cast<const uint32_t*>(comp);
    b.end_word = (in_off + in_len + 3u) >> 2;
    b.widx = in_off >> 2;
    b.cbase = b.widx;
    b.load_cache(lane);
    b.buf = 0;
    b.cnt = 0;
    b.refill(lane);
    b.drop((int)(in_off & 3u) * 8);
    uint32_t A = out_off;                 // absolute address of the next output byte
    const uint32_t A_end = out_off + out_len;
    unsigned err = kInflateOk;
    bool final_seen = false;
    PROF_DECL;
    BT_DECL;
#ifdef FTK_INFLATE_PROFILE
    int lane_dep = lane;
    PROF(7, lane_dep);
    prof_acc[7] = 0;
#endif

    // write-behind: granule g = absolute [g * kGran, (g + 1) * kGran), clipped to this block's range
    auto flush = [&](uint32_t g) {
        const uint32_t a0 = g << kGranShift;
        if (a0 >= out_off && a0 + kGran <= A_end) {
#pragma unroll
            for (int h = 0; h < kGran / 1024; ++h) {
                const uint32_t a = a0 + (uint32_t)h * 1024u + (uint32_t)lane * 16u;
                *reinterpret_cast<uint4*>(out + a) = *reinterpret_cast<const uint4*>(&L.ring[a & kRingMask]);
            }
        } else {
            for (uint32_t a = a0 + (uint32_t)lane; a < a0 + kGran; a += 64)
                if (a >= out_off && a < A_end) out[a] = L.ring[a & kRingMask];
        }
    };

    while (!final_seen && err == kInflateOk) {
        BTIME(2);
        b.refill(lane);
        final_seen = b.take(1) != 0;
        const unsigned type = b.take(2);
        if (type == 0) {  // stored: to the byte boundary, LEN, NLEN, LEN bytes
            b.drop(b.cnt & 7);
            b.refill(lane);
            const unsigned len = b.take(16);
            b.refill(lane);
            const unsigned nlen = b.take(16);
            if ((len ^ nlen) != 0xffffu) { err = kInflateBadStored; break; }
            if (A + len > A_end) { err = kInflateOverrun; break; }
            for (unsigned k = 0; k < len; ++k) {
                b.refill(lane);
                const unsigned v = b.take(8);
                if (lane == 0) L.ring[A & kRingMask] = (uint8_t)v;
                ++A;
                if ((A & (kGran - 1)) == 0) flush((A >> kGranShift) - 1);
            }
            continue;
        }
        if (type == 3) { err = kInflateBadBlockType; break; }
        int hlit = 288, hdist = 30;
        if (type == 1) {  // fixed code (RFC 1951 3.2.6)
            for (int i = lane; i < 288; i += 64) L.lens[i] = i < 144 ? 8 : i < 256 ? 9 : i < 280 ? 7 : 8;
            if (lane < 30) L.lens[288 + lane] = 5;
        } else {          // dynamic code: the code lengths themselves are Huffman coded (3.2.7)
            hlit = (int)b.take(5) + 257;
            hdist = (int)b.take(5) + 1;
            const int hclen = (int)b.take(4) + 4;
            if (hlit > 286 || hdist > 30) { err = kInflateBadLengths; break; }
            if (lane < 19) L.lens[320 + lane] = 0;
            for (int i = 0; i < hclen; ++i) {
                b.refill(lane);
                const unsigned v = b.take(3);
                // order of the code-length code lengths: 16 17 18 0 8 7 9 6 10 5 11 4 12 3 13 2 14 1 15
                const unsigned long long order = 0xf1e2d3c4b5a69780ull;  // entries 3..18, one nibble each, lowest first
                const int sym = i < 3 ? 16 + i : (int)((order >> (4 * (i - 3))) & 15ull);
                if (lane == 0) L.lens[320 + sym] = (uint8_t)v;
            }
            if (!build_table(L, &L.lens[320], 19, kPreRoot, L.pre, 0, lane)) { err = kInflateBadLengths; break; }
            const int total = hlit + hdist;
            int i = 0, prev = 0;
            while (i < total && err == kInflateOk) {
                b.refill(lane);
                const int e = UNI(L.pre[b.peek(kPreRoot)]);
                const int l = e & 15, s = e >> 4;
                if (l == 0) { err = kInflateBadSymbol; break; }
                b.drop(l);
                if (s < 16) {
                    if (lane == 0) L.lens[i] = (uint8_t)s;
                    prev = s;
                    ++i;
                    continue;
                }
                int rep, val = 0;
                if (s == 16) {
                    if (i == 0) { err = kInflateBadLengths; break; }
                    rep = 3 + (int)b.take(2);
                    val = prev;
                } else if (s == 17) {
                    rep = 3 + (int)b.take(3);
                } else {
                    rep = 11 + (int)b.take(7);
                }
                if (i + rep > total) { err = kInflateBadLengths; break; }
                for (int k = lane; k < rep; k += 64) L.lens[i + k] = (uint8_t)val;
                i += rep;
                prev = val;
            }
            if (err != kInflateOk) break;
            if (UNI(L.lens[256]) == 0) { err = kInflateBadLengths; break; }
            // distance lengths behind the literal / length ones, as build_table wants them: contiguous
            if (hlit != 288)
                for (int k = lane; k < hdist; k += 64) L.lens[288 + k] = L.lens[hlit + k];
        }
        if (!build_table(L, L.lens, hlit, kLitRoot, reinterpret_cast<uint16_t*>(L.pair), 0, lane) ||
            !build_table(L, &L.lens[288], hdist, kDistRoot, reinterpret_cast<uint16_t*>(L.dist), 1, lane)) {
            err = kInflateBadLengths;
            break;
        }
        build_pairs(L, lane);
        build_dist(L, lane);
        BTIME(0);  // the block's header and tables
        // ---- the symbols of this DEFLATE block --------------------------------------------------------
        // Literals: lane k looks up the symbol that would start k bits ahead (one LDS gather); the wave then hops from
        // symbol to symbol through that register - one v_readlane and a handful of scalar operations per hop, the
        // hops marked in two 64-bit masks (on the chain / a pair of literals) - for as long as the symbols are
        // literals and their root bits lie inside the valid part of the bit buffer.  Then the marked lanes write
        // their own one or two bytes to the ring, at the rank the masks give them (v_mbcnt): no per-literal vector
        // work, no pending-literal register.
        auto advance = [&](uint32_t n) {  // n bytes were written at A
            const uint32_t A0 = A;
            A += n;
            if ((A >> kGranShift) != (A0 >> kGranShift)) flush(A0 >> kGranShift);
        };
#if FTK_INFLATE_WINDOWED
        uint32_t bp = b.bitpos();  // the windows work from the absolute bit position; `b` follows when the serial path needs it
        bool b_moved = false;
        bool lanes_rest = false;   // LANES: the symbol at bp stopped a super-window - one window of the older kind takes it
#endif
        for (;;) {
            PROF(7, lane_dep);  // (whatever ran since the last mark: the serial path, the loop's bookkeeping)
            BTIME(2);  // (windows of the older kind, the serial path)
#if FTK_INFLATE_WINDOWED
            if (LANES && lane_tok != nullptr && !lanes_rest) {
                if (b_moved) {
                    bp = b.bitpos();
                    b_moved = false;
                }
                // ---- (A) every lane decodes its own stretch of kLaneBits bits from the stretch's first bit ----
                const uint32_t in_bits = b.end_word * 32u;
                // one symbol at bit position `at` of this lane: bits it takes (0: the tables do not resolve it), its token
                // a code longer than a table's root, bit by bit, by the lane itself (decode_long's walk): symbol or -1
                auto long_code = [&](unsigned v, int which, int& len_out) -> int {
                    int code = 0, first = 0, index = 0;
                    for (int len = 1; len <= 15; ++len) {
                        code |= (int)(v & 1u);
                        v >>= 1;
                        const int count = L.cnt[which][len];
                        if (code - count < first) {
                            len_out = len;
                            return L.sorted[which * 288 + index + (code - first)];
                        }
                        index += count;
                        first = (first + count) << 1;
                        code <<= 1;
                    }
                    return -1;
                };
                LTIME_DECL;
                // A lane reads its stretch front to back, so its input is a WINDOW IN REGISTERS: the word its position is in
                // and the four behind it (r0..r3, nx), moved up as the position crosses words (a symbol takes at most 48
                // bits: one or two words a trip) and refilled by the lane's own loads - which have a trip or more to arrive,
                // since a symbol is decoded from r0..r2 alone.  The first version staged the super-window's words
                // in LDS (4.3 KB a wave, the largest single item of the 16 KB that kept a CU at ten waves) and every symbol
                // began with an LDS round trip for its three words.  Words behind the payload read as zeros.
                struct InWin {
                    uint32_t wi;
                    unsigned r0, r1, r2, r3, nx;
                };
                auto ldw = [&](uint32_t wi) -> unsigned { return wi < b.end_word ? b.w[wi] : 0u; };
                auto win_open = [&](InWin& W, uint32_t at) {
                    W.wi = at >> 5;
                    W.r0 = ldw(W.wi);
                    W.r1 = ldw(W.wi + 1u);
                    W.r2 = ldw(W.wi + 2u);
                    W.r3 = ldw(W.wi + 3u);
                    W.nx = ldw(W.wi + 4u);
                };
                auto win_seek = [&](InWin& W, uint32_t at) {  // (at most two words further than the window's first)
                    const uint32_t adv = (at >> 5) - W.wi;
                    if (adv != 0u) {
                        const bool one = adv == 1u;
                        W.r0 = one ? W.r1 : W.r2;
                        W.r1 = one ? W.r2 : W.r3;
                        W.r2 = one ? W.r3 : W.nx;
                        if (one) {
                            W.r3 = W.nx;
                            W.nx = ldw(W.wi + 5u);
                        } else {
                            W.r3 = ldw(W.wi + 5u);
                            W.nx = ldw(W.wi + 6u);
                        }
                        W.wi += adv;
                    }
                };
                auto decode_at = [&](const InWin& W, uint32_t at, unsigned& nb, unsigned& tok) {
                    const unsigned c_lo = W.r0, c_mid = W.r1, c_hi = W.r2;  // (W.wi == at >> 5: the caller has moved the window)
                    const unsigned sh = at & 31u;
                    const unsigned w0 = __builtin_amdgcn_alignbit(c_mid, c_lo, sh), w1 = __builtin_amdgcn_alignbit(c_hi, c_mid, sh);
                    const unsigned E = L.pair[w0 & ((1u << kLitRoot) - 1u)];
                    unsigned k1 = (E >> 5) & 3u, lb = E & 31u, lbase = (E >> 8) & 511u, xb = (E >> 20) & 7u;
                    unsigned lit = (((E >> 8) & 0xffu) << 8) | (((E >> 20) & 0xffu) << 16);
                    if (E == 0u) {  // (rare: a literal / length code longer than the root)
                        int len = 0;
                        const int sym = long_code(w0, 0, len);
                        lb = (unsigned)len;
                        k1 = 0u;
                        lbase = 0u;  // (stays 0 for end of block and for no code at all: the lane stops)
                        xb = 0u;
                        if (sym >= 0 && sym < 256) {
                            k1 = 1u;
                            lit = (unsigned)sym << 8;
                        } else if (sym > 256) {
                            const unsigned ls = (unsigned)sym - 257u;
                            if (ls < 8u) lbase = ls + 3u;
                            else if (ls == 28u) lbase = 258u;
                            else if (ls < 28u) { xb = (ls >> 2) - 1u; lbase = ((4u + (ls & 3u)) << xb) + 3u; }
                            else lbase = 511u;
                        }
                    }
                    const unsigned wl = __builtin_amdgcn_alignbit(w1, w0, lb);
                    const unsigned mlen = lbase + (wl & ((1u << xb) - 1u));
                    const unsigned wd = wl >> xb;
                    unsigned D = L.dist[wd & ((1u << kDistRoot) - 1u)];
                    if (D == 0u && k1 == 0u && lbase != 0u && lbase != 511u) {  // (rarer still: a long distance code)
                        int len = 0;
                        const int ds = long_code(wd, 1, len);
                        if (ds >= 0) D = dist_entry((unsigned)len, (unsigned)ds);
                    }
                    const unsigned dbits = D & 15u, dxb = (D >> 4) & 15u, dbase = D >> 8;
                    const unsigned mdist = dbase + ((wd >> dbits) & ((1u << dxb) - 1u));
                    const bool is_match = k1 == 0u && lbase != 0u && lbase != 511u && D != 0u && dbase != 0x7fffffu;
                    nb = is_match ? lb + xb + dbits + dxb : (k1 ? lb : 0u);
                    // token: kind (1 / 2 literals, 3 match) | literal bytes at bits 8 and 16, or length << 2 | distance << 11
                    tok = is_match ? (3u | (mlen << 2) | (mdist << 11)) : (k1 | lit);
                };
                const uint32_t p0 = bp + (uint32_t)lane * (uint32_t)kLaneBits, sub_end = p0 + (uint32_t)kLaneBits;
                // token k of lane l at [k][l]: the lanes run in lock step, so a trip's 64 tokens are one 256-byte store (lane
                // by lane - [l][k] - every trip wrote to 64 cache lines, and the address unit was busy with little else)
                uint32_t* spec = lane_tok + lane;
                uint32_t* catchup = lane_tok + 64 * kLaneTok + lane;
                // What a lane remembers of the positions it took for symbol starts, in LDS: a CHECKPOINT per 32-bit word of the
                // stretch - the first start inside the word (5 bits, + 1 so that 0 says none) and the number of the token that
                // starts there, 16 bits at [word][lane].  A chain that has met the chain of (A) walks its starts from there
                // on, so it is recognised at the first start of the NEXT word at the latest, and the tokens it wrote in
                // between are the same tokens: nothing is lost but a few trips.  (The first version kept a mask of every
                // bit: twice the LDS, and the mask had to be counted through to number the token a lane had landed on.)
                uint16_t* const cp16 = reinterpret_cast<uint16_t*>(lanes_vis);
#pragma unroll
                for (int k = 0; k < kLaneStartWords / 64; ++k) lanes_vis[k * 64 + lane] = 0u;
                unsigned last_word = ~0u;  // the word of the stretch the last checkpoint was written for
                uint32_t pos = p0;
                int ntok = 0;
                bool stopped = false;               // the chain ended at something a window of the older kind must take
                bool active = p0 < in_bits;
                if (!active) stopped = true;        // (a lane behind the payload: nothing of it counts)
                LSTAT(0, 1);
                InWin W;
                win_open(W, p0);
                while (__ballot(active)) {
                    LACC(l_trips_a, 1);
                    if (active) {
                        unsigned nb, tok;
                        decode_at(W, pos, nb, tok);
                        if (nb == 0u || ntok == kLaneTok) {
                            stopped = true;
                            active = false;
                        } else {
                            const unsigned rel = pos - p0, word = rel >> 5;
                            if (word != last_word) cp16[word * 64u + (unsigned)lane] = (uint16_t)(((unsigned)ntok << 6) | ((rel & 31u) + 1u));
                            last_word = word;
                            pos += nb;
                            active = pos < sub_end;
                            // (the window's loads BEFORE the token's store: memory operations are counted off in order, so a
                            // load behind the store would not be seen to arrive before the store is acknowledged)
                            if (active) win_seek(W, pos);
                            spec[64 * ntok++] = tok;
                        }
                    }
                }
                LTIME(10);  // staging + pass A
                // ---- (B) true starts: lane l's first symbol starts where lane l - 1's chain ends ----
                const uint32_t spec_end = pos;
                const bool spec_stop = stopped;
                uint32_t my_end = spec_end, cur_start = p0;
                bool my_stop = spec_stop;
                int first_valid = 0, ncatch = 0;    // valid tokens: catchup[0, ncatch) then spec[first_valid, ntok)
                uint64_t unsettled = 0;
                for (int round = 0; round < kLaneRounds; ++round) {
                    const uint32_t prev_end = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)my_end, 0x138, 0xf, 0xf, false);  // wave_shr:1
                    const uint32_t c = lane == 0 ? bp : prev_end;
                    const uint64_t stops = __ballot(my_stop);
                    const bool behind_stop = (stops & ((1ull << lane) - 1ull)) != 0ull;  // a lane in front has stopped: nothing behind it counts
                    bool need = lane > 0 && !behind_stop && c != cur_start;
                    unsettled = __ballot(need);
                    if (!unsettled) break;
                    uint32_t q = c;
                    int nc = 0;
                    bool joined = false, cstop = false, go = need;
                    int join_tok = 0;  // the number of the token of (A) the lane landed on
                    if (need) win_open(W, c);
                    LACC(l_rounds, 1);
                    if (round == 1) LACC(l_need2, __popcll(unsettled));
                    while (__ballot(go)) {
                        LACC(l_trips_b, 1);
                        if (round == 0) LACC(l_trips_b1, 1);
                        if (go) {
                            const uint32_t rel = q - p0;  // (c >= p0: the lane in front ran to the end of its stretch or beyond)
                            // (the symbol at q is decoded whether or not q turns out to be a visited position: the checkpoint
                            // and the table entries then come in ONE LDS round trip, not one behind the other)
                            const bool inside = rel < (uint32_t)kLaneBits;
                            const unsigned word = (inside ? rel : 0u) >> 5;
                            const unsigned visw = cp16[word * 64u + (unsigned)lane];
                            unsigned nb, tok;
                            decode_at(W, q, nb, tok);
                            if (inside && (visw & 63u) == (rel & 31u) + 1u) {
                                joined = true;
                                join_tok = (int)(visw >> 6);
                                go = false;
                            } else if (q >= sub_end) {
                                go = false;
                            } else {
                                if (nb == 0u || nc == kLaneCatch) {
                                    cstop = true;
                                    go = false;
                                } else {
                                    q += nb;
                                    win_seek(W, q);
                                    catchup[64 * nc++] = tok;
                                }
                            }
                        }
                    }
                    if (round == 0) LACC(l_nojoin1, __popcll(__ballot(need && !joined)));
                    if (need) {
                        cur_start = c;
                        ncatch = nc;
                        if (joined) {
                            first_valid = join_tok;
                            my_end = spec_end;
                            my_stop = spec_stop;
                        } else {
                            first_valid = ntok;
                            my_end = q;
                            my_stop = cstop;
                        }
                    }
                }
                // lanes that count: in front of (and including) the first one that stopped, in front of the first one whose
                // start was still moving when the rounds ran out
                {
                    const uint32_t prev_end = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)my_end, 0x138, 0xf, 0xf, false);
                    const uint32_t c = lane == 0 ? bp : prev_end;
                    unsettled = __ballot(lane > 0 && c != cur_start);
                }
                const uint64_t stops = __ballot(my_stop);
                int n_lanes = 64;
                if (stops) n_lanes = min(n_lanes, __ffsll((unsigned long long)stops));          // the stopped lane itself counts
                if (unsettled) n_lanes = min(n_lanes, __ffsll((unsigned long long)unsettled) - 1);
                LTIME(11);  // the rounds
                LSTAT(1, n_lanes);
                LSTAT(7, stops ? 1 : 0);
                LSTAT(8, (unsettled && (!stops || __ffsll((unsigned long long)unsettled) - 1 < __ffsll((unsigned long long)stops))) ? 1 : 0);
                if (n_lanes >= 1) {
                    // ---- (C) where the valid tokens are: every lane keeps its count, its first valid token and its catch-up
                    // count in registers; fetch() finds the owner of the t-th token of the super-window and loads the token
                    // from the owner's column - no copy into one stream
                    const int nvalid = lane < n_lanes ? ncatch + (ntok - first_valid) : 0;
                    const int incl = wave_incl_scan(nvalid);
                    const int n_tokens = __builtin_amdgcn_readlane(incl, 63);
                    const int my_first = incl - nvalid;                  // tokens in front of this lane's
                    LSTAT(2, n_tokens);
                    lanes_vis[lane] = 0u;                                // (the slots fetch() drops owners in)
                    const uint32_t new_bp = (uint32_t)__builtin_amdgcn_readlane((int)my_end, n_lanes - 1);
                    const bool end_stop = ((stops >> (n_lanes - 1)) & 1ull) != 0ull;
                    // (tokens are written by one lane and read by another lane OF THE SAME WAVE: a workgroup-scope release /
                    // acquire pair.  The agent-scope pair of the first version made every super-window write back its
                    // XCD's L2 - buffer_wbl2 - and its token loads bypass the cache)
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    LTIME(12);
                    // The owners of tokens base .. base + 63.  The owner of `base` is the last lane with tokens that starts at
                    // or in front of it - one ballot; every lane whose tokens start further inside drops its number at the
                    // slot of its first token, and a running maximum hands it to the slots behind (the lanes' starts rise
                    // with their numbers); the owner's numbers come over ds_bpermute.  One LDS round trip where the
                    // bisection of the first version took six, one after the other.
                    auto fetch = [&](int base) -> unsigned {
                        const int t = base + lane;
                        const uint64_t le = __ballot(nvalid > 0 && my_first <= base);
                        const int o0 = le ? 63 - __clzll((long long)le) : 0;
                        const int rel = my_first - base;
                        if (nvalid > 0 && rel > 0 && rel < 64) lanes_vis[rel] = (uint32_t)lane;
                        unsigned own = lanes_vis[lane];
                        lanes_vis[lane] = 0u;
                        own = max(own, (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0x111, 0xf, 0xf, false));
                        own = max(own, (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0x112, 0xf, 0xf, false));
                        own = max(own, (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0x114, 0xf, 0xf, false));
                        own = max(own, (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0x118, 0xf, 0xf, false));
                        own = max(own, (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0x142, 0xa, 0xf, false));
                        own = max(own, (unsigned)__builtin_amdgcn_update_dpp(0, (int)own, 0x143, 0xc, 0xf, false));
                        const int lo = max((int)own, o0);
                        const int o_first = __builtin_amdgcn_ds_bpermute(lo << 2, my_first);
                        const int nc = __builtin_amdgcn_ds_bpermute(lo << 2, ncatch);
                        const int fv = __builtin_amdgcn_ds_bpermute(lo << 2, first_valid);
                        if (t >= n_tokens) return 0u;
                        const int loc = t - o_first;
                        const uint32_t* src = loc < nc ? lane_tok + 64 * kLaneTok + 64 * loc + lo : lane_tok + 64 * (fv + (loc - nc)) + lo;
                        return __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    };
                    // ---- (D) 64 tokens at a time: places from a prefix sum, literals stored, matches copied in order ----
                    bool fail_d = false;
                    LACC_DECL;
#pragma unroll
                    for (int i = 0; i < kLaneCap / 64; ++i) lanes_vis[192 + 64 * i + lane] = 0u;  // the rows the symbols drop their keys in
                    unsigned tok_next = fetch(0);
                    for (int base = 0; base < n_tokens && !fail_d;) {
                        const unsigned tok = tok_next;
#if defined(FTK_LANES_STATS) && FTK_LANES_STATS > 1
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        LTIME(16);  // waiting for the group's tokens
#endif
                        unsigned mark = tok & 3u;
                        const unsigned mlen = (tok >> 2) & 511u, mdist = tok >> 11;
                        const int olen = mark == 3u ? (int)mlen : (int)mark;
                        const int inc = wave_incl_scan(olen);
                        const uint32_t off = (uint32_t)(inc - olen);
                        const uint32_t room = min((uint32_t)kLaneCap, A_end - A);
                        const uint64_t bad = __ballot(mark != 0u && ((uint32_t)inc > room || (mark == 3u && mdist > (A - out_off) + off)));
                        uint32_t T;
                        int took = min(64, n_tokens - base);
                        if (bad) {
                            const int cut = __ffsll((unsigned long long)bad) - 1;
                            if (cut == 0) {
                                // the first token does not fit: it reaches beyond the block's data or in front of it
                                const bool far = __builtin_amdgcn_readlane((int)(mark == 3u && mdist > (A - out_off) + off), 0) != 0;
                                err = far ? kInflateBadDistance : kInflateOverrun;
                                fail_d = true;
                                break;
                            }
                            if (lane >= cut) mark = 0u;
                            took = cut;
                            T = (uint32_t)__builtin_amdgcn_readlane((int)off, cut);
                        } else {
                            T = (uint32_t)__builtin_amdgcn_readlane(inc, 63);
                        }
                        LTIME_D(17);  // places
                        tok_next = fetch(base + took);  // (on its way while this group's bytes are stored and copied)
                        LTIME_D(18);  // the next group's owners found
                        const uint64_t mm_any = __ballot(mark == 3u);
                        LACC(d_matches, __popcll(mm_any));
                        LACC(d_groups, 1);
#ifndef FTK_LANES_SKIP
#define FTK_LANES_SKIP 0   // (timing experiments only - results are wrong: 1 no byte resolution at all, 2 no doubling rounds, 4 no far loads)
#endif
                        if (mm_any && !(FTK_LANES_SKIP & 1)) {
                            // ---- the group's BYTES side by side (T <= kLaneCap = 384: six rows of 64).  Copying the matches in
                            // stream order costs an LDS read -> write round trip per match, ~30 of them per group of fragment
                            // rows, one after the other - four fifths of this loop's time when the chip is full.  Instead every
                            // symbol drops a key (its place << 17 | match << 16 | distance - 1 or its literal bytes) at its first
                            // output byte in a zeroed row of LDS; a running maximum over the rows hands every byte the key of
                            // its symbol (the place is the key's top, so the latest symbol at or in front of a byte wins); and
                            // then each byte knows where it comes from: a literal, a byte of the ring older than the group (one
                            // LDS read), a byte further back than the ring holds (one load from HBM - written behind long ago,
                            // see kFarDist), or an EARLIER BYTE OF THIS GROUP.  The last kind is a pointer; pointers are followed
                            // by doubling (state array in LDS: the value once known, else a pointer to a byte of the same value),
                            // ~log2 of the deepest chain of copies-of-copies rounds, and one pass stores the group.
                            // (keys and states share the rows: all keys are read before any state is written, the rounds
                            // start when all rows hold states, and the rows are zeroed again behind the group.  Every step
                            // is written row by row in loops of its own - all reads of a step, then all its writes - so that
                            // the rows' LDS round trips overlap instead of following each other.)
                            uint32_t* const S = lanes_vis + 192;      // (words 0..63 are the slots fetch() drops owners in)
                            constexpr unsigned kRes = 0x80000000u;
                            constexpr int kRows = kLaneCap / 64;
                            static_assert(192 + kRows * 64 <= kLaneLdsWords, "the byte states of a group must fit behind the owner slots");
                            if (mark) S[off] = (off << 17) | (mark == 3u ? (0x10000u | (mdist - 1u)) : ((tok >> 8) & 0xffffu));
                            const bool any_far = __ballot(mark == 3u && mdist > (unsigned)kFarDist) != 0ull;
                            // (the rows are worked on without asking row by row whether the group reaches them - a row behind
                            // its end holds zeros and resolves to nothing: the tests cost more than the rows, and kept the
                            // rows' instructions from being interleaved - in two sizes: half the rows for short groups)
                            auto rows = [&](auto nr_tag) {
                                constexpr int NR = decltype(nr_tag)::value;
                                unsigned stv[NR];
#pragma unroll
                                for (int i = 0; i < NR; ++i) stv[i] = S[lane + 64 * i];
#pragma unroll
                                for (int i = 0; i < NR; ++i) {
                                    unsigned k = stv[i];
                                    k = max(k, (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x111, 0xf, 0xf, false));
                                    k = max(k, (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x112, 0xf, 0xf, false));
                                    k = max(k, (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x114, 0xf, 0xf, false));
                                    k = max(k, (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x118, 0xf, 0xf, false));
                                    k = max(k, (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x142, 0xa, 0xf, false));
                                    k = max(k, (unsigned)__builtin_amdgcn_update_dpp(0, (int)k, 0x143, 0xc, 0xf, false));
                                    stv[i] = k;
                                }
                                unsigned carry = 0u;
#pragma unroll
                                for (int i = 0; i < NR; ++i) {
                                    stv[i] = max(stv[i], carry);
                                    carry = (unsigned)__builtin_amdgcn_readlane((int)stv[i], 63);
                                }
                                LTIME_D(21);  // keys dropped, read, running maximum
                                // where each byte comes from; the ring bytes of all rows are read together (a row that needs
                                // none reads its own slot)
                                unsigned old[NR];
#pragma unroll
                                for (int i = 0; i < NR; ++i) {
                                    const uint32_t j = (uint32_t)(lane + 64 * i);
                                    const unsigned k = stv[i];
                                    const uint32_t d = (k & 0x7fffu) + 1u;
                                    const bool from_ring = j < T && (k & 0x10000u) != 0u && j < d && d <= (uint32_t)kFarDist;
                                    old[i] = L.ring[(from_ring ? A + j - d : A + j) & kRingMask];
                                }
                                if (any_far && !(FTK_LANES_SKIP & 4)) {
                                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
#pragma unroll
                                    for (int i = 0; i < NR; ++i) {
                                        const uint32_t j = (uint32_t)(lane + 64 * i);
                                        const unsigned k = stv[i];
                                        const uint32_t d = (k & 0x7fffu) + 1u;
                                        if (j < T && (k & 0x10000u) != 0u && j < d && d > (uint32_t)kFarDist) old[i] = out[A + j - d];
                                    }
                                }
#pragma unroll
                                for (int i = 0; i < NR; ++i) {
                                    const uint32_t j = (uint32_t)(lane + 64 * i);
                                    const unsigned k = stv[i];
                                    const uint32_t rel = j - (k >> 17);
                                    const uint32_t d = (k & 0x7fffu) + 1u;
                                    unsigned st;
                                    if (j >= T) st = kRes;
                                    else if ((k & 0x10000u) == 0u) st = kRes | (((k & 0xffffu) >> (8u * (rel & 1u))) & 0xffu);
                                    else st = j >= d ? j - d : (kRes | old[i]);
                                    stv[i] = st;
                                    S[j] = st;
                                }
                                LTIME_D(22);  // sources and states
                                for (;;) {
                                    bool open = false;
#pragma unroll
                                    for (int i = 0; i < NR; ++i) open |= (stv[i] & kRes) == 0u;
                                    if (!__ballot(open) || (FTK_LANES_SKIP & 2)) break;
                                    LACC(d_rounds, 1);
                                    unsigned g[NR];
#pragma unroll
                                    for (int i = 0; i < NR; ++i) g[i] = S[(stv[i] & kRes) ? (unsigned)(lane + 64 * i) : stv[i]];
#pragma unroll
                                    for (int i = 0; i < NR; ++i) {
                                        if ((stv[i] & kRes) == 0u) {
                                            stv[i] = g[i];
                                            S[lane + 64 * i] = g[i];
                                        }
                                    }
                                }
                                LTIME_D(23);  // rounds of pointer doubling
#pragma unroll
                                for (int i = 0; i < NR; ++i) {
                                    const uint32_t j = (uint32_t)(lane + 64 * i);
                                    S[j] = 0u;
                                    if (j < T) L.ring[(A + j) & kRingMask] = (uint8_t)stv[i];
                                }
                            };
                            if (T <= 64u * (uint32_t)(kRows / 2)) rows(std::integral_constant<int, kRows / 2>{});
                            else rows(std::integral_constant<int, kRows>{});
                        } else if (mark == 1u || mark == 2u) {
                            const uint32_t at2 = A + off;
                            L.ring[at2 & kRingMask] = (uint8_t)(tok >> 8);
                            if (mark == 2u) L.ring[(at2 + 1u) & kRingMask] = (uint8_t)(tok >> 16);
                        }
                        LTIME_D(19);  // literals and matches
                        // (the next group's tokens are waited for HERE, in front of the write-behind: memory operations are
                        // counted off in order, so behind it the wait would also be for the granule's stores to be acknowledged)
                        asm volatile("" : "+v"(tok_next));
                        advance(T);
                        LTIME_D(20);  // write-behind
                        base += took;
                    }
                    if (fail_d) break;
                    LTIME(13);  // phase D
                    LSTAT(21, d_matches);
                    LSTAT(25, d_groups);
                    LSTAT(26, d_rounds);
                    LTIME_FLUSH;
                    bp = new_bp;
                    lanes_rest = end_stop;
                    BTIME(1);  // a super-window
                    continue;
                }
                lanes_rest = true;  // not even lane 0's stretch went through: a window of the older kind
            }
            lanes_rest = false;
            if (LANES) LSTAT(4, 1);
#endif
#if FTK_INFLATE_WINDOWED
            // ---- a window of symbols at once.  Every lane decodes, COMPLETELY and on the vector unit, the symbol that
            // would start at its bit offset: one or two literals, or a match with its extra bits and its distance (a
            // second gather, from the distance table, at the offset the length code ends at).  Each lane builds its
            // own 64 bits from three words of the input cache (ds_bpermute), so all 64 offsets are good and a window
            // consumes 64 bits or more - no scalar bit buffer, no refill.  What is left for the scalar unit - one per
            // CU, shared by all its waves, and what bounds this kernel - is the chain through the bit lengths:
            // v_readlane, add, test per symbol of ANY kind (six instructions a hop, three of them the scalar unit's).  Then: a prefix sum of the output lengths of the
            // lanes on the chain gives every symbol its place; the literal lanes store their bytes; the matches copy
            // in stream order (each may read what the one before it wrote).  A symbol the window cannot take - a code
            // longer than a table's root, end of block, a distance outside the block, output beyond kWinCap bytes or
            // the block's end - ends the window in front of it; when it is the first, the serial path below takes it.
            if (b_moved) {
                bp = b.bitpos();
                b_moved = false;
            }
            unsigned e;
            {
                if ((bp >> 5) - b.cbase > 59u) {  // the lanes need words (bp >> 5) .. ((bp + 63) >> 5) + 2 of the cache
                    b.cbase = bp >> 5;
                    b.load_cache(lane);
                }
                PROF(0, lane_dep);  // cache
                const uint32_t bo = bp + (uint32_t)lane;
                const int wi = (int)(((bo >> 5) - b.cbase) << 2);
                const unsigned c_lo = (unsigned)__builtin_amdgcn_ds_bpermute(wi, (int)b.cache);
                const unsigned c_mid = (unsigned)__builtin_amdgcn_ds_bpermute(wi + 4, (int)b.cache);
                const unsigned c_hi = (unsigned)__builtin_amdgcn_ds_bpermute(wi + 8, (int)b.cache);
                const unsigned sh = bo & 31u;
                const unsigned w0 = __builtin_amdgcn_alignbit(c_mid, c_lo, sh), w1 = __builtin_amdgcn_alignbit(c_hi, c_mid, sh);
                const unsigned E = L.pair[w0 & ((1u << kLitRoot) - 1u)];
                const unsigned k1 = (E >> 5) & 3u, lb = E & 31u, lbase = (E >> 8) & 511u, xb = (E >> 20) & 7u;
                const unsigned wl = __builtin_amdgcn_alignbit(w1, w0, lb);  // (behind the length code: 5 + 9 + 13 bits at most)
                const unsigned mlen = lbase + (wl & ((1u << xb) - 1u));
                const unsigned wd = wl >> xb;
                const unsigned D = L.dist[wd & ((1u << kDistRoot) - 1u)];
                const unsigned dbits = D & 15u, dxb = (D >> 4) & 15u, dbase = D >> 8;
                const unsigned mdist = dbase + ((wd >> dbits) & ((1u << dxb) - 1u));
                const bool is_match = k1 == 0u && lbase != 0u && lbase != 511u && D != 0u && dbase != 0x7fffffu;
                unsigned kind = is_match ? 3u : k1;  // 0: end of block, a long code, no such symbol
                // (a lane the chain must stop at says 64 bits: the add carries the position out of the window, one test
                // serves both ends of the loop, and the store of its kind - 0 - marks nothing)
                unsigned NB = kind ? (is_match ? lb + xb + dbits + dxb : lb) : 64u;
                PROF(1, NB);  // the lanes' bits, the two gathers and the decode
                // (round 5: the lanes on the chain are collected in a SCALAR bit mask - s_bitset1_b64, one scalar
                // instruction a hop - and every lane picks its kind from it once, behind the loop; until then each hop
                // marked its lane on the vector unit, v_cmp + v_cndmask = eight of a SIMD's cycles per symbol, and the
                // vector unit is what this kernel is short of.  Five instructions a hop, one of them the vector unit's.)
                unsigned mark, t;
                int pos;
                uint64_t chain;
                asm volatile(
                    "s_mov_b32 %[pos], 0\n\t"
                    "s_mov_b64 %[chain], 0\n"
                    "1:\n\t"
                    "v_readlane_b32 %[t], %[NB], %[pos]\n\t"
                    "s_bitset1_b64 %[chain], %[pos]\n\t"
                    "s_add_i32 %[pos], %[pos], %[t]\n\t"
                    "s_cmp_lt_u32 %[pos], 64\n\t"
                    "s_cbranch_scc1 1b\n\t"
                    "v_cndmask_b32 %[mark], 0, %[kind], %[chain]\n\t"
                    : [pos] "=&s"(pos), [mark] "=&v"(mark), [t] "=&s"(t), [chain] "=&s"(chain)
                    : [NB] "v"(NB), [kind] "v"(kind)
                    : "scc");
                PROF(2, mark);  // the chain
                pos = UNI(pos);
                if ((unsigned)UNI(t) == 64u) pos -= 64;  // stopped in front of a lane, not beyond the window
                const uint64_t on = __ballot(mark != 0u);
                if (on) {
                    const int olen = mark == 3u ? (int)mlen : (int)mark;
                    const int inc = wave_incl_scan(olen);
                    const uint32_t off = (uint32_t)(inc - olen);
                    const uint32_t room = min((uint32_t)kWinCap, A_end - A);
                    // the first symbol that does not fit or points outside the block ends the window in front of it
                    const uint64_t bad = __ballot(mark != 0u && ((uint32_t)inc > room || (mark == 3u && mdist > (A - out_off) + off)));
                    uint32_t T;
                    if (bad) {
                        const int cut = __ffsll((unsigned long long)bad) - 1;
                        if (lane >= cut) mark = 0u;
                        pos = cut;
                        T = (uint32_t)__builtin_amdgcn_readlane((int)off, cut);
                    } else {
                        T = (uint32_t)__builtin_amdgcn_readlane(inc, 63);
                    }
                    if (T) {
                        // ---- the window's bytes resolved side by side (round 4; launches that ask for it: VEC).  The serial match loop below costs one
                        // LDS read -> write round trip per match, one after the other (a third of a window's cycles, DESIGN
                        // 3.5).  When the window's output fits the wave (T <= 63 bytes) and every match reads the ring,
                        // lane j becomes output byte j instead: the chain's lanes push their number to the lane of their
                        // first output byte (ds_permute), a running maximum hands every byte its symbol, one gather brings
                        // the symbol's distance or literal bytes, and then each byte knows where it comes from - a literal,
                        // a ring byte written by an earlier window (one LDS read for all of them), or an EARLIER BYTE OF
                        // THIS WINDOW (in fragment rows the END column's digits copy the START column's, a few bytes back).
                        // The last kind is resolved by pointer jumping over the lanes (ds_bpermute: no memory), a round
                        // per doubling of the dependency depth - two rounds for rows - and one store writes the window.
                        if (VEC && T <= 63u && __ballot(mark == 3u) != 0ull && __ballot(mark == 3u && mdist > (unsigned)kFarDist) == 0ull) {
                            int own = __builtin_amdgcn_ds_permute((mark ? (int)off : 63) << 2, mark ? lane + 1 : 0);
                            own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x111, 0xf, 0xf, false));
                            own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x112, 0xf, 0xf, false));
                            own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x114, 0xf, 0xf, false));
                            own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x118, 0xf, 0xf, false));
                            own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x142, 0xa, 0xf, false));
                            own = max(own, __builtin_amdgcn_update_dpp(0, own, 0x143, 0xc, 0xf, false));
                            // what a byte needs of its symbol: where the symbol starts, its kind, its distance or bytes
                            const unsigned pk = (mark == 3u ? mdist : (((E >> 8) & 0xffu) | (((E >> 20) & 0xffu) << 8))) |
                                                (off << 16) | (mark << 22);
                            const unsigned oi = (unsigned)__builtin_amdgcn_ds_bpermute((own - 1) << 2, (int)pk);
                            const unsigned o_pay = oi & 0xffffu, rel = (unsigned)lane - ((oi >> 16) & 63u);
                            const bool is_m = (oi >> 22) == 3u;
                            const int src = lane - (int)o_pay;  // (matches) the window byte this one copies; < 0: an older byte
                            const unsigned old = L.ring[(A + (uint32_t)lane - o_pay) & kRingMask];
                            // state: 0x100 | value once known, else the lane to take it from
                            unsigned st = !is_m ? (0x100u | ((o_pay >> (8u * (rel & 1u))) & 0xffu))
                                                : (src < 0 ? (0x100u | old) : (unsigned)src);
                            if ((uint32_t)lane >= T) st = 0x100u;
                            while (__ballot((st & 0x100u) == 0u)) {
                                const unsigned g = (unsigned)__builtin_amdgcn_ds_bpermute((int)((st & 63u) << 2), (int)st);
                                if ((st & 0x100u) == 0u) st = g;
                            }
                            if ((uint32_t)lane < T) L.ring[(A + (uint32_t)lane) & kRingMask] = (uint8_t)st;
                            PROF(4, mark);
                            advance(T);
                            bp += (uint32_t)pos;
                            PROF(5, mark);
                            continue;
                        }
                        if (mark == 1u || mark == 2u) {
                            const uint32_t at = A + off;
                            L.ring[at & kRingMask] = (uint8_t)(E >> 8);
                            if (mark == 2u) L.ring[(at + 1u) & kRingMask] = (uint8_t)(E >> 20);
                        }
                        // the matches, in stream order; what kind of copy each needs was worked out by its lane
                        const unsigned ccase = (mdist >= mlen && mlen <= 64u && mdist <= (unsigned)kFarDist) ? 0u
                                               : mdist <= (unsigned)kFarDist                                ? 1u
                                                                                                            : 2u;
                        const unsigned lenc = mlen | (ccase << 16);
                        const uint32_t M0v = A + off;
                        PROF(3, mark);  // prefix sum, checks, literal stores
                        uint64_t mm = __ballot(mark == 3u);
                        if (mm && __ballot(mark == 3u && ccase != 0u) == 0ull) {
                            // Every match of the window is of the common kind (one step, LDS to LDS): the loop without
                            // the kind test and without touching EXEC - the lanes beyond a match's length read and
                            // write a spare LDS byte instead (the code-length array, idle while symbols are decoded).
                            // 17 instructions a match, five of them the scalar unit's (find-first, clear, wait, test,
                            // branch); the loop below spends nine there.
                            unsigned lc, md, m0, va, vs, vx;
                            int l;
                            const unsigned spare = (unsigned)offsetof(WaveLds, lens);
                            asm volatile(
                                "1:\n\t"
                                "s_ff1_i32_b64 %[l], %[mm]\n\t"
                                "s_bitset0_b64 %[mm], %[l]\n\t"
                                "v_readlane_b32 %[lc], %[lenc], %[l]\n\t"
                                "v_readlane_b32 %[md], %[mdist], %[l]\n\t"
                                "v_readlane_b32 %[m0], %[M0v], %[l]\n\t"
                                "v_cmp_gt_u32 vcc, %[lc], %[lane]\n\t"
                                "v_add_u32 %[va], %[m0], %[lane]\n\t"
                                "v_subrev_u32 %[vs], %[md], %[va]\n\t"
                                "v_and_b32 %[vs], %[mask], %[vs]\n\t"
                                "v_and_b32 %[va], %[mask], %[va]\n\t"
                                "v_cndmask_b32 %[vs], %[spare], %[vs], vcc\n\t"
                                "v_cndmask_b32 %[va], %[spare], %[va], vcc\n\t"
                                "ds_read_u8 %[vx], %[vs]\n\t"
                                "s_waitcnt lgkmcnt(0)\n\t"
                                "ds_write_b8 %[va], %[vx]\n\t"
                                "s_cmp_lg_u64 %[mm], 0\n\t"
                                "s_cbranch_scc1 1b\n\t"
                                : [mm] "+s"(mm), [lc] "=&s"(lc), [md] "=&s"(md), [m0] "=&s"(m0), [l] "=&s"(l), [va] "=&v"(va),
                                  [vs] "=&v"(vs), [vx] "=&v"(vx)
                                : [lenc] "v"(lenc), [mdist] "v"(mdist), [M0v] "v"(M0v), [lane] "v"(lane), [mask] "i"(kRingMask),
                                  [spare] "v"(spare)  // (a VGPR: with VCC the instruction may read no other scalar)
                                : "vcc", "scc", "memory");
                        } else if (mm) {
                            for (;;) {
                                // The common matches (one step, LDS to LDS) in stream order, written out: 19 instructions
                                // each (find-first, clear, three v_readlane, test, lane mask, two addresses, read, wait,
                                // write, loop) where the compiler's version of the same loop takes ~35.  It stops at a
                                // match of another kind - handed to the C++ below, already taken off the mask - with
                                // st = 1 (more matches behind it) or 2 (none), or with st = 0 when the mask is empty.
                                unsigned lc, md, m0, st, va, vs, vx;
                                int l;
                                uint64_t sv;
                                asm volatile(
                                    "s_mov_b64 %[sv], exec\n"
                                    "1:\n\t"
                                    "s_ff1_i32_b64 %[l], %[mm]\n\t"
                                    "s_bitset0_b64 %[mm], %[l]\n\t"
                                    "v_readlane_b32 %[lc], %[lenc], %[l]\n\t"
                                    "v_readlane_b32 %[md], %[mdist], %[l]\n\t"
                                    "v_readlane_b32 %[m0], %[M0v], %[l]\n\t"
                                    "s_cmp_ge_u32 %[lc], 0x10000\n\t"
                                    "s_cbranch_scc1 3f\n\t"
                                    "v_cmp_gt_u32 vcc, %[lc], %[lane]\n\t"
                                    "s_mov_b64 exec, vcc\n\t"
                                    "v_add_u32 %[va], %[m0], %[lane]\n\t"
                                    "v_subrev_u32 %[vs], %[md], %[va]\n\t"
                                    "v_and_b32 %[vs], %[mask], %[vs]\n\t"
                                    "ds_read_u8 %[vx], %[vs]\n\t"
                                    "v_and_b32 %[va], %[mask], %[va]\n\t"
                                    "s_waitcnt lgkmcnt(0)\n\t"
                                    "ds_write_b8 %[va], %[vx]\n\t"
                                    "s_mov_b64 exec, %[sv]\n\t"
                                    "s_cmp_lg_u64 %[mm], 0\n\t"
                                    "s_cbranch_scc1 1b\n\t"
                                    "s_mov_b32 %[st], 0\n\t"
                                    "s_branch 4f\n"
                                    "3:\n\t"
                                    "s_cmp_lg_u64 %[mm], 0\n\t"
                                    "s_cselect_b32 %[st], 1, 2\n"
                                    "4:\n\t"
                                    : [mm] "+s"(mm), [lc] "=&s"(lc), [md] "=&s"(md), [m0] "=&s"(m0), [st] "=&s"(st), [l] "=&s"(l),
                                      [sv] "=&s"(sv), [va] "=&v"(va), [vs] "=&v"(vs), [vx] "=&v"(vx)
                                    : [lenc] "v"(lenc), [mdist] "v"(mdist), [M0v] "v"(M0v), [lane] "v"(lane), [mask] "i"(kRingMask)
                                    : "vcc", "scc", "memory");
                                st = (unsigned)UNI(st);
                                if (st == 0u) break;
                                const int len = (int)((unsigned)UNI(lc) & 0xffffu), d = UNI(md);
                                const uint32_t M0 = (uint32_t)UNI(m0);
                                if (((unsigned)UNI(lc) >> 16) == 1u) {
                                    // LDS to LDS, period d (see the serial path)
                                    int done = 0, DD = d;
                                    while (done < len) {
                                        const int n = min(min(len - done, DD), 64);
                                        if (lane < n) {
                                            const uint32_t a = M0 + (uint32_t)(done + lane);
                                            L.ring[a & kRingMask] = L.ring[(a - (uint32_t)DD) & kRingMask];
                                        }
                                        done += n;
                                        if (2 * DD <= done + d) DD *= 2;
                                    }
                                } else {
                                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                                    for (int o = lane; o < len; o += 64) {
                                        const uint32_t a = M0 + (uint32_t)o;
                                        L.ring[a & kRingMask] = out[a - (uint32_t)d];
                                    }
                                }
                                if (st == 2u) break;
                            }
                        }
                        PROF(4, mark);  // the matches' copies
                        advance(T);
                        bp += (uint32_t)pos;
                        PROF(5, mark);  // write-behind
                        continue;
                    }
                }
                e = (unsigned)__builtin_amdgcn_readlane((int)E, 0);
                PROF(6, mark);  // a window that was not taken
            }
            b.seek(bp, lane);
            b_moved = true;
            if (e & 0x60u) {  // (cannot happen unless the block is out of room: the serial path reports it) literal(s) at the front
                const uint32_t n = (e >> 5) & 3u;
                if (A + n > A_end) { err = kInflateOverrun; break; }
                if (lane == 0) {
                    L.ring[A & kRingMask] = (uint8_t)(e >> 8);
                    if (n == 2u) L.ring[(A + 1u) & kRingMask] = (uint8_t)(e >> 20);
                }
                advance(n);
                b.drop((int)(e & 31u));
                continue;
            }
#else
            b.refill(lane);
            const unsigned E = L.pair[(unsigned)(b.buf >> lane) & ((1u << kLitRoot) - 1u)];
            const int limit = b.cnt - kLitRoot;
            int pos = 0;
            // The hops are marked ON THE VECTOR UNIT: the lane whose bit offset a hop lands on keeps the entry's kind
            // (1: one literal, 2: a pair).  The scalar unit - one per CU, shared by every wave of it, and what bounds
            // this kernel - is left with the chain itself (entry -> test -> length -> position); the two masks come
            // out of two ballots after the loop instead of four 64-bit shift / or instructions per hop.
            // The loop is written out: as C++ the compiler keeps a 64-bit "which exit" flag alive across the back
            // edge (s_mov_b64 / s_cselect_b64 / s_andn2_b64 per hop).  Nine instructions per hop: four scalar, two
            // branches, three vector.  (No manual wait states are needed: the lane select of v_readlane is written
            // by the scalar unit, v_cmp -> v_cndmask through VCC is interlocked.)
            const unsigned kindv = (E >> 5) & 3u;  // every lane's own entry kind: the lane a hop lands on keeps it
            unsigned mark, e;
            int t;
            asm volatile(
                "s_mov_b32 %[pos], 0\n\t"
                "v_mov_b32 %[mark], 0\n"
                "1:\n\t"
                "v_readlane_b32 %[e], %[E], %[pos]\n\t"
                "s_and_b32 %[t], %[e], 0x60\n\t"
                "s_cbranch_scc0 2f\n\t"
                "v_cmp_eq_u32 vcc, %[pos], %[lane]\n\t"
                "v_cndmask_b32 %[mark], %[mark], %[kindv], vcc\n\t"
                "s_and_b32 %[t], %[e], 31\n\t"
                "s_add_i32 %[pos], %[pos], %[t]\n\t"
                "s_cmp_le_i32 %[pos], %[limit]\n\t"
                "s_cbranch_scc1 1b\n"
                "2:\n\t"
                : [pos] "=&s"(pos), [mark] "=&v"(mark), [e] "=&s"(e), [t] "=&s"(t)
                : [E] "v"(E), [lane] "v"(lane), [kindv] "v"(kindv), [limit] "s"(limit)
                : "vcc", "scc");
            const uint64_t on_chain = __ballot(mark != 0u), pairs = __ballot(mark == 2u);
            if (on_chain) {
                const uint32_t total = (uint32_t)(__popcll(on_chain) + __popcll(pairs));
                if (A + total > A_end) { err = kInflateOverrun; break; }
                const unsigned before = __builtin_amdgcn_mbcnt_hi((unsigned)(on_chain >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)on_chain, 0u)) +
                                        __builtin_amdgcn_mbcnt_hi((unsigned)(pairs >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pairs, 0u));
                if ((on_chain >> lane) & 1ull) {
                    const uint32_t at = A + before;
                    L.ring[at & kRingMask] = (uint8_t)(E >> 8);
                    if ((pairs >> lane) & 1ull) L.ring[(at + 1u) & kRingMask] = (uint8_t)(E >> 20);
                }
                advance(total);
                b.drop(pos);
            }
            if (e & 0x60u) continue;  // out of valid bits, not of literals
#endif
            b.refill(lane);  // the symbol at the front is no literal: its code and extra bits take up to 20
            unsigned le = e;  // length entry: bits | base << 8 | extra bits << 20 (base 0: end of block)
            if (e) {
                b.drop((int)(e & 31u));
            } else {
                const int sym = decode_long(L, b, 0);  // a code longer than the root: bit by bit
                if (sym < 0) { err = kInflateBadSymbol; break; }
                if (sym < 256) {
                    if (A + 1u > A_end) { err = kInflateOverrun; break; }
                    if (lane == 0) L.ring[A & kRingMask] = (uint8_t)sym;
                    advance(1u);
                    continue;
                }
                const unsigned ls = (unsigned)sym - 257u;
                unsigned base = 0, eb = 0;
                if (sym == 256) base = 0;
                else if (ls < 8u) base = ls + 3u;
                else if (ls == 28u) base = 258u;
                else if (ls < 28u) { eb = (ls >> 2) - 1u; base = ((4u + (ls & 3u)) << eb) + 3u; }
                else base = 511u;
                le = (base << 8) | (eb << 20);
            }
            const unsigned lbase = (le >> 8) & 511u;
            if (lbase == 0) break;  // end of block
            if (lbase == 511u) { err = kInflateBadSymbol; break; }
            const int len = (int)(lbase + b.take((int)(le >> 20)));
            b.refill(lane);
            unsigned de = (unsigned)UNI(L.dist[b.peek(kDistRoot)]);
            if (de) {
                b.drop((int)(de & 15u));
            } else {
                const int ds = decode_long(L, b, 1);
                if (ds < 0) { err = kInflateBadSymbol; break; }
                de = dist_entry(0u, (unsigned)ds);
            }
            if ((de >> 8) == 0x7fffffu) { err = kInflateBadSymbol; break; }
            const int d = (int)((de >> 8) + b.take((int)((de >> 4) & 15u)));
            if ((uint32_t)d > A - out_off) { err = kInflateBadDistance; break; }
            if (A + (uint32_t)len > A_end) { err = kInflateOverrun; break; }
            const uint32_t A0 = A;
            if (d >= len && len <= 64 && d <= kFarDist) {
                // the common case (a row repeats part of an earlier one): one step, sources all older than the match
                if (lane < len) {
                    const uint32_t a = A0 + (uint32_t)lane;
                    L.ring[a & kRingMask] = L.ring[(a - (uint32_t)d) & kRingMask];
                }
            } else if (d <= kFarDist) {
                // LDS to LDS.  The pattern has period d: after `done` bytes, [A0 - d, A0 + done) is valid, so a
                // step may copy up to D bytes from D back for any multiple D of d with D <= done + d; D doubles.
                int done = 0, D = d;
                while (done < len) {
                    const int n = min(min(len - done, D), 64);
                    if (lane < n) {
                        const uint32_t a = A0 + (uint32_t)(done + lane);
                        L.ring[a & kRingMask] = L.ring[(a - (uint32_t)D) & kRingMask];
                    }
                    done += n;
                    if (2 * D <= done + d) D *= 2;
                }
            } else {
                // further back than kFarDist: those bytes are in HBM already (see the header comment)
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
                for (int o = lane; o < len; o += 64) {
                    const uint32_t a = A0 + (uint32_t)o;
                    L.ring[a & kRingMask] = out[a - (uint32_t)d];
                }
            }
            A = A0 + (uint32_t)len;
            if ((A >> kGranShift) != (A0 >> kGranShift)) flush(A0 >> kGranShift);
        }
    }
#ifdef FTK_INFLATE_PROFILE
    PROF(7, lane_dep);
    if (lane == 0)
        for (int i = 0; i < 8; ++i) {
            atomicAdd(&g_prof[i], prof_acc[i]);
            atomicAdd(&g_prof[8 + i], prof_n[i]);
        }
#endif
    if (err == kInflateOk && A != A_end) err = kInflateShort;
    // the unfinished granule
    if ((A & (kGran - 1)) != 0 || A == out_off) flush(A >> kGranShift);
    BTIME(2);
    BT_FLUSH;
#ifdef FTK_INFLATE_TIMING
    if (lane == 0 && blk < 65536) g_block_ticks[2 * blk + 1] = wall_clock64();
#endif
    if (err != kInflateOk && lane == 0) {
        if (atomicAdd(&status->n_bad, 1u) == 0) {
            status->first_bad = (unsigned)blk;
            status->reason = err;
        }
    }
    if (LANES && lane_slot >= 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (lane == 0) atomicExch(&lane_scratch->busy[lane_slot], 0u);
    }
}

// ---- CRC-32 of every block's data (the gzip trailer carries the expected value; the host compares) ----------
// One wave per block: every lane takes a contiguous stripe of ceil(len / 64) bytes (bit-serial, on the vector unit), then the 64 stripe CRCs are folded pairwise with
// crc(A || B) = crc(A) * x^(8 |B|) mod P  xor  crc(B)   (polynomial arithmetic in the reflected representation).
__device__ __forceinline__ uint32_t gf2_mul(uint32_t a, uint32_t b) {  // a * b mod P
    uint32_t p = 0;
    for (uint32_t m = 1u << 31; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ 0xedb88320u : b >> 1;
    }
    return p;
}

// x^(8 * 2^k) mod P for k = 0..16 (a block holds at most 2^16 bytes), worked out by the compiler: the fold's powers
// are products of these - no squaring chain at run time (the chain was 15 x 2 multiplications of ~190 operations per
// level of the fold: as much work as the bit-serial stripes themselves)
constexpr uint32_t gf2_mul_c(uint32_t a, uint32_t b) {
    uint32_t p = 0;
    for (uint32_t m = 1u << 31; m; m >>= 1) {
        if (a & m) p ^= b;
        b = (b & 1u) ? (b >> 1) ^ 0xedb88320u : b >> 1;
    }
    return p;
}
struct Pow8Table {
    uint32_t v[17];
    constexpr Pow8Table() : v{} {
        uint32_t sq = 0x00800000u;  // x^8
        for (int k = 0; k < 17; ++k) {
            v[k] = sq;
            sq = gf2_mul_c(sq, sq);
        }
    }
};
constexpr Pow8Table kPow8{};

__device__ __forceinline__ uint32_t gf2_x_pow_8n(uint32_t n) {  // x^(8 n) mod P, n < 2^17
    uint32_t p = 1u << 31;  // x^0
#pragma unroll
    for (int k = 0; k < 17; ++k)
        if ((n >> k) & 1u) p = gf2_mul(kPow8.v[k], p);
    return p;
}

// Four waves per workgroup share four 256-entry tables in LDS (slicing-by-4: one 32-bit word of data per four look-ups;
// the tables are built by the workgroup itself, 8 bit-steps per entry); each wave takes BGZF blocks of its own.  The
// bit-serial stripes this replaces spent 32 vector operations per byte (1.1 TB/s with the table-driven fold); a lane's
// look-ups go to random entries, so a wave's ds_read meets bank conflicts (256 words over 32 banks) - still a fraction of
// the bit-serial cost.
constexpr int kCrcWaves = 4;

__global__ __launch_bounds__(64 * kCrcWaves) void bgzf_crc_kernel(const InflateBlock* __restrict__ tab, int n_blocks,
                                                                  const uint8_t* __restrict__ out, uint32_t* __restrict__ crc_out) {
    __shared__ uint32_t T[4][256];
    {
        const int i = threadIdx.x;  // 256 threads: one entry of each table
        uint32_t c = (uint32_t)i;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xedb88320u & (0u - (c & 1u)));
        T[0][i] = c;
        __syncthreads();
        uint32_t t1 = (c >> 8) ^ T[0][c & 255u];
        uint32_t t2 = (t1 >> 8) ^ T[0][t1 & 255u];
        uint32_t t3 = (t2 >> 8) ^ T[0][t2 & 255u];
        T[1][i] = t1;
        T[2][i] = t2;
        T[3][i] = t3;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int blk = blockIdx.x * kCrcWaves + wv; blk < n_blocks; blk += gridDim.x * kCrcWaves) {
        const uint32_t len = tab[blk].out_len;
        const uint8_t* p = out + tab[blk].out_off;
        // 64 stripes of `stripe` bytes cut from the END of the block's data: lane 63 ends where the data ends, every
        // stripe to the right of a non-empty one is full, the ragged (or empty) ones are the first.  The fold's
        // multiplier - x^(8 * bytes to the right) - is then the same for every lane of a level, x^(8 * d * stripe):
        // one squaring per level instead of a power per lane.  (Cut from the start, as the first version did, the
        // ragged stripe was on the right and every level paid for two different powers.)
        const uint32_t stripe = max((((len + 63u) / 64u) + 15u) & ~15u, 16u);
        const long long hi = (long long)len - (long long)(63 - lane) * stripe, lo = hi - stripe;
        const uint32_t a = (uint32_t)max(lo, 0ll), e = (uint32_t)max(hi, 0ll);
        uint32_t c = 0xffffffffu;
        auto byte_step = [&](uint32_t i) { c = T[0][(c ^ p[i]) & 255u] ^ (c >> 8); };
        uint32_t i = a;
        // one aligned 16-byte load per 16 bytes (with byte loads every 128-byte line was fetched 128 times and a launch
        // of 9 000 blocks ran at a third of the rate of one of 1 900: the lines do not stay in a CU's L1)
        auto word_step = [&](uint32_t w) {
            const uint32_t x = c ^ w;
            c = T[3][x & 255u] ^ T[2][(x >> 8) & 255u] ^ T[1][(x >> 16) & 255u] ^ T[0][x >> 24];
        };
        for (; i < e && ((reinterpret_cast<uintptr_t>(p) + i) & 15u); ++i) byte_step(i);
        // 128 bytes - a cache line's worth - per lane at a time: the lanes' stripes lie a kilobyte apart, so a wave's
        // load touches 64 lines; taking a line in one burst of eight loads keeps it from being fetched eight times
        // (a chip-filling launch has 64 MB of such lines in flight, more than the L2s hold)
        for (; i + 128u <= e; i += 128u) {
            uint4 v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = *reinterpret_cast<const uint4*>(p + i + 16 * q);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                word_step(v[q].x);
                word_step(v[q].y);
                word_step(v[q].z);
                word_step(v[q].w);
            }
        }
        for (; i + 16u <= e; i += 16u) {
            const uint4 v = *reinterpret_cast<const uint4*>(p + i);
            word_step(v.x);
            word_step(v.y);
            word_step(v.z);
            word_step(v.w);
        }
        for (; i < e; ++i) byte_step(i);
        bool have = e > a;
        c = have ? ~c : 0u;  // CRC of the stripe (0 for an empty one: the neutral element of the fold)
        uint32_t pw = gf2_x_pow_8n(stripe);
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t c_r = __shfl_down(c, d, 64);
            const bool have_r = __shfl_down((int)have, d, 64) != 0;
            if ((lane & (2 * d - 1)) == 0 && have_r) {  // crc(A || B) = crc(A) * x^(8 |B|) xor crc(B); |B| = d stripes
                c = have ? (gf2_mul(pw, c) ^ c_r) : c_r;
                have = true;
            }
            pw = gf2_mul(pw, pw);
        }
        if (lane == 0) crc_out[blk] = c;
    }
}

}  // namespace

#ifdef FTK_INFLATE_PROFILE
extern "C" int ftk_debug_inflate_profile(unsigned long long* out, int reset) {  // out[0..8): cycles, out[8..16): passes
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_prof), sizeof(unsigned long long) * 16);
    if (reset) {
        unsigned long long z[32] = {};
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_prof), z, sizeof z);
    }
    return rc;
}
#endif
#ifdef FTK_LANES_STATS
extern "C" int ftk_debug_lanes_stats(unsigned long long* out, int reset) {
    int rc = (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lanes_stats), sizeof(unsigned long long) * 32);
    if (reset) {
        unsigned long long z[32] = {};
        rc |= (int)hipMemcpyToSymbol(HIP_SYMBOL(g_lanes_stats), z, sizeof z);
    }
    return rc;
}
#endif
#ifdef FTK_INFLATE_TIMING
extern "C" int ftk_debug_inflate_ticks(unsigned long long* out, int n_blocks) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_block_ticks), sizeof(unsigned long long) * 2 * (size_t)n_blocks);
}
#endif

// Scratch of the lane-parallel symbol loop: kLaneSlots token regions and their busy words (~370 MB), one allocation per
// device, made (and cleared) the first time a launch asks for it.  A failed allocation is NOT remembered - that launch
// takes the windowed loop and the next one asks again -, and inflate_release_scratch (ftk_cache_trim) gives the memory
// back: several ranks that share a GPU, or a host that keeps the library loaded between jobs, need not hold it.
namespace {
std::mutex g_lane_mu;
std::map<int, LaneScratch*> g_lane_have;
}  // namespace

static LaneScratch* lane_scratch_of_device() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    std::lock_guard<std::mutex> lk(g_lane_mu);
    auto it = g_lane_have.find(dev);
    if (it != g_lane_have.end()) return it->second;
    LaneScratch* p = nullptr;
    const size_t bytes = sizeof(unsigned) * kLaneSlots + sizeof(uint32_t) * kLaneSlots * kLaneSlotWords + 256;
    if (hipMalloc((void**)&p, bytes) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    if (hipMemset(p, 0, sizeof(unsigned) * kLaneSlots) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(p);
        return nullptr;
    }
    g_lane_have[dev] = p;
    return p;
}

size_t inflate_release_scratch() {
    std::lock_guard<std::mutex> lk(g_lane_mu);
    size_t freed = 0;
    int cur = 0;
    const bool have_cur = hipGetDevice(&cur) == hipSuccess;
    for (auto& kv : g_lane_have) {
        if (!kv.second) continue;
        if (hipSetDevice(kv.first) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {  // (no launch may still use it)
            (void)hipGetLastError();
            continue;
        }
        if (hipFree(kv.second) == hipSuccess) freed += sizeof(unsigned) * kLaneSlots + sizeof(uint32_t) * kLaneSlots * kLaneSlotWords + 256;
        kv.second = nullptr;
    }
    for (auto it = g_lane_have.begin(); it != g_lane_have.end();) it = it->second ? std::next(it) : g_lane_have.erase(it);
    if (have_cur) (void)hipSetDevice(cur);
    return freed;
}

void inflate_launch(hipStream_t s, const uint8_t* d_comp, const InflateBlock* d_tab, int n_blocks, uint8_t* d_out,
                    InflateStatus* d_status, uint32_t* d_crc, bool vector_matches) {
    if (n_blocks <= 0) return;
#ifdef FTK_INFLATE_VECMATCH
    vector_matches = FTK_INFLATE_VECMATCH != 0;
#endif
    // Which symbol loop: the lane-parallel one unless FTK_INFLATE_LANES=0 asks for the windowed loop (read per launch: the
    // tests hold both against zlib).  Since (D) resolves a group's bytes side by side it is ahead at every launch size
    // (a 1 883-block launch 1.2 vs 3.4 ms, chip-filling text 3.4 vs 7.5 ms, BAM records 5.0 vs 9.6 ms).
    const char* le = getenv("FTK_INFLATE_LANES");
    const bool lanes = !(le && *le) || atoi(le) != 0;
    LaneScratch* ls = lanes ? lane_scratch_of_device() : nullptr;
    if (ls)
        hipLaunchKernelGGL((bgzf_inflate_kernel<false, true>), dim3(n_blocks), dim3(64), 0, s, d_comp, d_tab, n_blocks, d_out, d_status, ls);
    else if (vector_matches)
        hipLaunchKernelGGL((bgzf_inflate_kernel<true, false>), dim3(n_blocks), dim3(64), 0, s, d_comp, d_tab, n_blocks, d_out, d_status, ls);
    else
        hipLaunchKernelGGL((bgzf_inflate_kernel<false, false>), dim3(n_blocks), dim3(64), 0, s, d_comp, d_tab, n_blocks, d_out, d_status, ls);
    if (d_crc) {
        // (workgroups of four waves, a block per wave)
        const int groups = std::min((n_blocks + kCrcWaves - 1) / kCrcWaves, 1 << 20);
        hipLaunchKernelGGL(bgzf_crc_kernel, dim3(groups), dim3(64 * kCrcWaves), 0, s, d_tab, n_blocks, d_out, d_crc);
    }
}

}  // namespace ftk
