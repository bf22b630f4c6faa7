// Device-side row parser of the streaming text decoder: see ftk_textparse.h.
//
// Per piece of inflated text (~190 MB, ~7 M rows): a set-up launch (device-inflated pieces: carry, range, scan state
// cleared) and ONE pass that finds the lines and parses them (lines_rows_kernel: line ends per 4 KB block, the block's
// place among all lines by decoupled look-back, one thread per line: name span, three unsigned decimals, strand, line
// end -> columns).  Lines that are anything but plain rows are counted (the host then parses the piece itself); lines
// whose contig name differs from the previous line's are listed (contig runs).  Algorithmic bytes: the text once plus
// 10 B per row out.  Until round 3 this was four kernels (count, scan, positions, rows) over a line index in HBM.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "ftk_textparse.h"

namespace ftk {
namespace {

constexpr int kT = 256;  // threads per block of the newline kernels
constexpr int kB = 16;   // bytes per thread
static_assert(kT * kB == kTextBlockBytes, "block geometry");

// The range a kernel works on: its arguments, or (device-inflated pieces) the one the set-up kernel recorded in the
// summary.  The range then starts at any byte: the kernels work from the 16-byte boundary below it (`lead` bytes
// earlier) and ignore line ends in front of the first line.
struct Range {
    const uint8_t* text;
    size_t n;
    unsigned lead;
};
__device__ __forceinline__ Range range_of(const uint8_t* text, size_t n, const TextSummary* ind) {
    if (!ind) return {text, n, 0u};
    const unsigned off = ind->text_off, lead = off & 15u;
    return {text + (off - lead), ind->text_len ? (size_t)ind->text_len + lead : 0, lead};
}

// line ends among the 16 bytes at `off` (bit j = byte off + j); the bytes themselves come back in `v` (zeros beyond n)
__device__ __forceinline__ unsigned nl_mask16(const uint8_t* __restrict__ text, size_t off, size_t n, uint4& v) {
    unsigned m = 0;
    if (off + kB <= n) {
        v = *reinterpret_cast<const uint4*>(text + off);
    } else {
        unsigned w[4] = {0u, 0u, 0u, 0u};
        for (int j = 0; j < kB && off + j < n; ++j) w[j >> 2] |= (unsigned)text[off + j] << (8 * (j & 3));
        v = make_uint4(w[0], w[1], w[2], w[3]);
    }
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (((w[k] >> (8 * j)) & 0xffu) == (unsigned)'\n') m |= 1u << (4 * k + j);
    if (off + kB > n) m &= n > off ? (1u << (n - off)) - 1u : 0u;
    return m;
}

// inclusive scan within a wave
__device__ __forceinline__ unsigned wave_incl_scan(unsigned v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned t = __shfl_up(v, d, 64);
        if (lane >= d) v += t;
    }
    return v;
}

// ---- the scan state of the single-pass line finder (decoupled look-back) ------------------------------------------
// state[b] of logical block b: flag << 32 | value; flag 0 = nothing yet, 1 = value is the block's own line-end count,
// 2 = value is the count of all blocks up to and including b.  One 64-bit word, written and read with one agent-scope
// atomic each, so flag and value always belong together.  state[n_max_blocks] is the ticket counter: a block's
// logical number is the order in which it STARTED, so every predecessor a block waits for is already running.
constexpr unsigned long long kAgg = 1ull << 32, kIncl = 2ull << 32;
__device__ __forceinline__ unsigned long long st_load(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_store(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 1..10 decimal digits ending in `term` before the line end `e` (text[e] is the '\n'); p moves past `term`
// Where a line's bytes are read from: the piece in HBM, or the block's copy of its 4 KB (and the bytes in front of them)
// in LDS - a row is read byte by byte in dependent steps, and a step on LDS takes a tenth of one on L2.
struct GlobalText {
    const uint8_t* __restrict__ p;
    __device__ __forceinline__ uint8_t operator[](uint32_t i) const { return p[i]; }
};
struct LdsText {
    const uint8_t* p;  // the LDS copy
    uint32_t base;     // position of its first byte in the range
    __device__ __forceinline__ uint8_t operator[](uint32_t i) const { return p[i - base]; }
};

template <typename Text>
__device__ __forceinline__ bool dev_digits(const Text& t, uint32_t& p, uint32_t e, uint8_t term, unsigned long long& v) {
    const uint32_t s = p;
    unsigned long long x = 0;
    while (p < e) {
        const unsigned d = (unsigned)t[p] - (unsigned)'0';
        if (d > 9) break;
        x = x * 10 + d;
        ++p;
    }
    if (p == s || p - s > 10 || p >= e || t[p] != term) return false;
    ++p;
    v = x;
    return true;
}

// The same row the host's one-pass parser accepts (plain_row in ftk_decode.cpp), with any contig name: line i of the
// piece spans t[b, e) (t[e] is its '\n'), the line before it starts at pb.  Returns false for anything but a plain row.
template <typename Text>
__device__ __forceinline__ bool parse_row(const Text& t, size_t i, uint32_t b, uint32_t e, uint32_t pb, bool last,
                                          size_t max_lines, int bed6, int32_t* __restrict__ o_start, int32_t* __restrict__ o_end,
                                          uint8_t* __restrict__ o_mapq, uint8_t* __restrict__ o_strand,
                                          TextSummary* __restrict__ sum, int indirect) {
    uint32_t p = b;
    while (p < e && t[p] != '\t') ++p;
    const uint32_t name_len = p - b;
    unsigned long long fs = 0, fe = 0, mq = 0;
    uint8_t strand = 0;
    bool ok = name_len > 0 && p < e && t[b] != '#';
    if (ok) {
        ++p;
        ok = dev_digits(t, p, e, '\t', fs) && dev_digits(t, p, e, '\t', fe);
    }
    if (ok && bed6) {  // column 3 (a name) is not read
        while (p < e && t[p] != '\t') ++p;
        ok = p < e;
        ++p;
    }
    if (ok) ok = dev_digits(t, p, e, '\t', mq);
    if (ok) {
        ok = p < e;
        if (ok) {
            strand = t[p];
            ok = strand != '\t' && strand != '\r';
            ++p;
            if (p < e && t[p] == '\r') ++p;
            ok = ok && p == e;
        }
    }
    // rows the host would skip (coordinates beyond int32) also send the piece to the host parser
    ok = ok && fs <= 0x7fffffffull && fe <= 0x7fffffffull;
    if (ok && i < max_lines) {
        o_start[i] = (int32_t)fs;
        o_end[i] = (int32_t)fe;
        o_mapq[i] = (uint8_t)(mq < 255 ? mq : 255);
        o_strand[i] = strand == '+' ? 1 : 0;
        bool new_run = i == 0;
        if (!new_run) {
            const uint32_t pe = b - 1;
            new_run = pb + name_len >= pe || t[pb + name_len] != '\t';
            for (uint32_t k = 0; !new_run && k < name_len; ++k) new_run = t[pb + k] != t[b + k];
        }
        if (new_run) {
            const unsigned slot = atomicAdd(&sum->n_runs, 1u);
            if (slot < (unsigned)kTextMaxRuns) {
                sum->run_line[slot] = (unsigned)i;
                sum->run_off[slot] = b;
            }
            if (indirect) {  // the host has no copy of this text: hand it the name
                if (slot < (unsigned)kTextNamedRuns && name_len < (uint32_t)kTextNameBytes) {
                    for (uint32_t k = 0; k < name_len; ++k) sum->run_name[slot][k] = t[b + k];
                    sum->run_name[slot][name_len] = 0;
                } else {
                    sum->name_overflow = 1;
                }
            }
        }
    }
    if (!ok && last) sum->last_line_bad = 1;
    return ok;
}

// start of the line that ends in front of position `from` (from - 1 is its last byte or `from` its '\n'): the byte behind
// the nearest '\n' below `from`, or `lead` (the first line of the range starts there)
template <typename Text>
__device__ __forceinline__ uint32_t line_start_before(const Text& t, uint32_t from, uint32_t lead) {
    uint32_t p = from;
    while (p > lead && t[p - 1] != '\n') --p;
    return p;
}

// Lines found AND parsed in one pass over a piece (round 4; before: count, scan, positions and rows as four launches,
// the text read three times and a 4-byte line index written and read back).  A block takes 4 KB of text into LDS (and
// the kHalo bytes in front of them): line ends per 16 bytes, block scan, its place among all lines by decoupled
// look-back over the blocks in front (state[], see above), the positions of its line ends in LDS - and then one thread
// per line that ENDS in the block parses it from the LDS copy: its start is the previous line end (for the block's
// first line: found by one wave in the halo), its row index the block's base plus its rank.  A line that starts in
// front of the halo is read from HBM instead.  The total goes to sum->n_lines (last block); a piece with more lines
// than the outputs hold is reported (overflow) and its rows beyond them are not written.
constexpr int kHalo = 256;
// Line ends a block keeps in LDS.  A plain row is at least 10 bytes ("1\t1\t2\t0\t+\n"), so a 4 KB block of plain rows
// holds at most 409; a block with more line ends than this holds rows the device does not parse anyway, and says so
// (overflow: the host's field-rule parser takes the piece).  The cap is what keeps the kernel at 5.4 KB of LDS per
// block: the inflate kernel's waves hold 150 of a CU's 160 KB while they are resident, and a block that needs 12.5 KB
// (the first version: a slot for every byte of the block) waits for two of them to retire before it can start.
constexpr int kMaxLinesPerBlock = 512;
__global__ __launch_bounds__(kT) void lines_rows_kernel(const uint8_t* __restrict__ text_, size_t n_,
                                                         unsigned long long* __restrict__ state, unsigned n_max_blocks,
                                                         size_t max_lines, int bed6, int32_t* __restrict__ o_start,
                                                         int32_t* __restrict__ o_end, uint8_t* __restrict__ o_mapq,
                                                         uint8_t* __restrict__ o_strand, TextSummary* __restrict__ sum,
                                                         int indirect) {
    __shared__ unsigned s_bid, s_base, s_first, s_prev;
    __shared__ unsigned wave_tot[kT / 64];
    __shared__ unsigned short nlpos[kMaxLinesPerBlock];  // offsets of the block's line ends within the block
    __shared__ __align__(16) uint8_t s_text[kHalo + kTextBlockBytes];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_bid = atomicAdd(reinterpret_cast<unsigned*>(state + n_max_blocks), 1u);
    __syncthreads();
    const unsigned bid = s_bid;
    const Range R = range_of(text_, n_, indirect ? sum : nullptr);
    const uint8_t* __restrict__ text = R.text;
    const size_t n = R.n;
    const unsigned n_blocks = (unsigned)((n + kTextBlockBytes - 1) / kTextBlockBytes);
    if (bid >= n_blocks) {
        if (bid == 0 && tid == 0) sum->n_lines = 0;  // (an empty range)
        return;
    }
    const size_t blk_off = (size_t)bid * kTextBlockBytes;
    const size_t off = blk_off + (size_t)tid * kB;
    uint4 mine = make_uint4(0u, 0u, 0u, 0u);
    unsigned m = off < n ? nl_mask16(text, off, n, mine) : 0;
    *reinterpret_cast<uint4*>(&s_text[kHalo + tid * kB]) = mine;
    if (tid < kHalo / kB && bid > 0)  // (block 0 has nothing in front of it)
        *reinterpret_cast<uint4*>(&s_text[tid * kB]) = *reinterpret_cast<const uint4*>(text + blk_off - kHalo + (size_t)tid * kB);
    if (off == 0) m &= ~0u << R.lead;
    const unsigned c = __popc(m);
    const unsigned incl = wave_incl_scan(c, lane);
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    unsigned before = 0, total = 0;
#pragma unroll
    for (int k = 0; k < kT / 64; ++k) {
        if (k < wv) before += wave_tot[k];
        total += wave_tot[k];
    }
    {
        unsigned idx = before + incl - c;
        while (m) {
            const int j = __ffs(m) - 1;
            m &= m - 1;
            if (idx < (unsigned)kMaxLinesPerBlock) nlpos[idx] = (unsigned short)(tid * kB + j);
            ++idx;
        }
    }
    if (wv == 0) {  // this block's place among all lines
        unsigned base = 0;
        if (bid > 0) {
            if (lane == 0) st_store(state + bid, kAgg | total);
            int j = (int)bid - 1;
            for (;;) {
                const int idx = j - lane;
                unsigned long long v = kIncl;  // (in front of block 0: nothing)
                if (idx >= 0)
                    do v = st_load(state + idx);
                    while ((v >> 32) == 0ull);
                const unsigned long long done = __ballot((v >> 32) == 2ull);
                const int stop = done ? __ffsll(done) - 1 : 63;  // nearest predecessor that already knows its inclusive count
                unsigned part = lane <= stop ? (unsigned)v : 0u;
#pragma unroll
                for (int d = 32; d > 0; d >>= 1) part += __shfl_down(part, d, 64);
                base += (unsigned)__shfl(part, 0, 64);
                if (done) break;
                j -= 64;
            }
        }
        if (lane == 0) {
            st_store(state + bid, kIncl | (unsigned long long)(base + total));
            s_base = base;
            if (bid == n_blocks - 1) {
                sum->n_lines = (unsigned long long)base + total;
                if ((size_t)base + total > max_lines) sum->overflow = 1;
            }
        }
    }
    __syncthreads();
    if (total == 0) return;
    if (total > (unsigned)kMaxLinesPerBlock) {  // not a block of plain rows (see kMaxLinesPerBlock)
        if (tid == 0) sum->overflow = 1;
        return;
    }
    // positions below are absolute (in the range); the LDS copy holds [lds_lo, blk_off + 4 KB)
    const uint32_t lds_lo = bid > 0 ? (uint32_t)blk_off - (uint32_t)kHalo : 0u;
    const LdsText LT{s_text, (uint32_t)blk_off - (uint32_t)kHalo};  // (block 0: positions start at kHalo of the copy)
    const GlobalText GT{text};
    if (wv == 0) {
        // Where the block's first line starts, and the line in front of it (for its run test): the two line ends in
        // front of the block's first one.  The wave looks at the 128 bytes in front of it at once (rows are ~26 bytes;
        // one thread walking back byte by byte - a dependent load each - held 255 others at the barrier for ~50 loads
        // per block and tripled the kernel's time); what lies further back is walked serially, in HBM.
        const uint32_t e0 = (uint32_t)blk_off + nlpos[0];
        const uint32_t span = e0 - max(R.lead, lds_lo);  // bytes in front of e0 that are both in the range and in the copy
        const bool all_seen = lds_lo <= R.lead;          // the copy reaches back to the start of the range
        const bool v0 = (uint32_t)lane < span, v1 = (uint32_t)lane + 64u < span;
        const unsigned long long m0 = __ballot(v0 && LT[e0 - 1u - (uint32_t)lane] == '\n');
        const unsigned long long m1 = __ballot(v1 && LT[e0 - 65u - (uint32_t)lane] == '\n');
        if (lane == 0) {
            // k-th byte back (k = 0: e0 - 1) is a line end -> the line behind it starts at e0 - k
            const uint32_t seen = min(span, 128u);  // bytes looked at
            uint32_t f, pv;
            const int k1 = m0 ? __ffsll(m0) - 1 : (m1 ? 64 + __ffsll(m1) - 1 : -1);
            if (k1 >= 0) f = e0 - (uint32_t)k1;
            else f = (all_seen && span <= 128u) ? R.lead : line_start_before(GT, e0 - seen, R.lead);
            if (f == R.lead) {
                pv = f;
            } else {
                int k2 = -1;
                if (k1 >= 0) {
                    const unsigned long long r0 = k1 < 63 ? m0 & (~0ull << (k1 + 1)) : 0ull;
                    const unsigned long long r1 = k1 < 64 ? m1 : (k1 < 127 ? m1 & (~0ull << (k1 - 63)) : 0ull);
                    k2 = r0 ? __ffsll(r0) - 1 : (r1 ? 64 + __ffsll(r1) - 1 : -1);
                }
                if (k2 >= 0) pv = e0 - (uint32_t)k2;
                else if (k1 >= 0 && all_seen && span <= 128u) pv = R.lead;
                else pv = line_start_before(GT, k1 >= 0 ? e0 - seen : f - 1u, R.lead);
            }
            s_first = f;
            s_prev = pv;
        }
    }
    __syncthreads();
    const unsigned base = s_base;
    const uint32_t last_e = (uint32_t)(n - 1);  // the range ends with a line end: that line is the piece's last
    for (unsigned k0 = 0; k0 < total; k0 += kT) {
        const unsigned k = k0 + (unsigned)tid;
        bool bad = false;
        if (k < total) {
            const uint32_t e = (uint32_t)blk_off + nlpos[k];
            const uint32_t b = k > 0 ? (uint32_t)blk_off + nlpos[k - 1] + 1u : s_first;
            const uint32_t pb = k > 1 ? (uint32_t)blk_off + nlpos[k - 2] + 1u : (k == 1 ? s_first : s_prev);
            const size_t i = (size_t)base + k;
            if (min(b, pb) >= lds_lo)  // the line and the one in front of it lie in the LDS copy
                bad = !parse_row(LT, i, b, e, pb, e == last_e, max_lines, bed6, o_start, o_end, o_mapq, o_strand, sum, indirect);
            else
                bad = !parse_row(GT, i, b, e, pb, e == last_e, max_lines, bed6, o_start, o_end, o_mapq, o_strand, sum, indirect);
        }
        const unsigned long long mask = __ballot(bad);
        if (lane == 0 && mask) atomicAdd(&sum->n_bad, (unsigned long long)__popcll(mask));
    }
}

// Device-inflated pieces: carry, skip, last line end (see textparse_launch_inflated).  One block.
__global__ __launch_bounds__(256) void piece_setup_kernel(uint8_t* __restrict__ text, uint32_t data_off, uint32_t data_len,
                                                          const uint8_t* __restrict__ prev_text,
                                                          const TextSummary* __restrict__ prev_sum, uint32_t first_skip,
                                                          int eof, TextSummary* __restrict__ sum,
                                                          unsigned long long* __restrict__ state, unsigned n_state) {
    __shared__ unsigned best;
    const int tid = threadIdx.x;
    if (blockIdx.x > 0) {  // the other blocks clear the line finder's scan state (and its ticket) for this piece
        for (size_t k = (size_t)(blockIdx.x - 1) * 256 + tid; k < n_state; k += (size_t)(gridDim.x - 1) * 256) state[k] = 0ull;
        return;
    }
    unsigned tail_prev = 0, prev_end = 0;
    if (prev_sum) {
        tail_prev = prev_sum->tail_len;
        prev_end = prev_sum->text_off + prev_sum->text_len;
    }
    if (tail_prev > kTextCarryMax || tail_prev > data_off) {  // (the previous piece already said carry_overflow)
        if (tid == 0) sum->carry_overflow = 1;
        tail_prev = 0;
    }
    for (unsigned k = tid; k < tail_prev; k += 256) text[data_off - tail_prev + k] = prev_text[prev_end + k];
    const unsigned off = data_off - tail_prev + (prev_sum ? 0u : min(first_skip, data_len));
    unsigned end = data_off + data_len;
    if (tid == 0) best = 0;  // position behind the last '\n' in [off, end), 0 = none
    __syncthreads();
    // the last line end lies within kTextCarryMax of the end, or the unfinished line is too long to carry
    const unsigned lo = end - off > kTextCarryMax ? end - kTextCarryMax : off;
    unsigned mine = 0;
    for (unsigned p = lo + tid; p < end; p += 256)
        if (text[p] == '\n') mine = p + 1;
    atomicMax(&best, mine);
    __syncthreads();
    if (tid == 0) {
        unsigned last = best;
        if (eof && end > off && last != end) {  // the file's last row has no line end: give it one (the buffer has the room)
            text[end] = '\n';
            last = ++end;
        }
        if (last == 0) {
            sum->text_off = off;
            sum->text_len = 0;
            sum->tail_len = end - off;
            if (end - off > kTextCarryMax) sum->carry_overflow = 1;
        } else {
            sum->text_off = off;
            sum->text_len = last - off;
            sum->tail_len = end - last;
        }
    }
}

// One thread per row, grid-stride (a run starts at any row of the piece and lands at any row of the block: element
// accesses, 256 + 256 + 64 + 64 bytes per wave-instruction).  Streaming data: read once, written once.
template <bool BAM>
__global__ __launch_bounds__(256) void append_rows_kernel(int32_t* __restrict__ ds, int32_t* __restrict__ de,
                                                          uint8_t* __restrict__ dq, uint8_t* __restrict__ dt,
                                                          int32_t* __restrict__ da, int32_t* __restrict__ db,
                                                          const int32_t* __restrict__ ss, const int32_t* __restrict__ se,
                                                          const uint8_t* __restrict__ sq, const uint8_t* __restrict__ st,
                                                          const int32_t* __restrict__ sa, const int32_t* __restrict__ sb,
                                                          size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        ds[i] = __builtin_nontemporal_load(ss + i);
        de[i] = __builtin_nontemporal_load(se + i);
        dq[i] = __builtin_nontemporal_load(sq + i);
        dt[i] = __builtin_nontemporal_load(st + i);
        if (BAM) {
            da[i] = __builtin_nontemporal_load(sa + i);
            db[i] = __builtin_nontemporal_load(sb + i);
        }
    }
}

}  // namespace

void append_rows_launch(hipStream_t s, int32_t* dst_start, int32_t* dst_end, uint8_t* dst_mapq, uint8_t* dst_strand,
                        int32_t* dst_r1s, int32_t* dst_r1e, const int32_t* src_start, const int32_t* src_end,
                        const uint8_t* src_mapq, const uint8_t* src_strand, const int32_t* src_r1s, const int32_t* src_r1e,
                        size_t n) {
    if (!n) return;
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 4096);
    if (dst_r1s)
        hipLaunchKernelGGL(append_rows_kernel<true>, dim3(blocks), dim3(256), 0, s, dst_start, dst_end, dst_mapq, dst_strand,
                           dst_r1s, dst_r1e, src_start, src_end, src_mapq, src_strand, src_r1s, src_r1e, n);
    else
        hipLaunchKernelGGL(append_rows_kernel<false>, dim3(blocks), dim3(256), 0, s, dst_start, dst_end, dst_mapq, dst_strand,
                           dst_r1s, dst_r1e, src_start, src_end, src_mapq, src_strand, src_r1s, src_r1e, n);
}

size_t textparse_scratch_bytes(size_t text_bytes) {
    return ((text_bytes + kTextCarryMax + 64) / kTextBlockBytes + 4) * sizeof(unsigned long long);
}

void textparse_launch_inflated(hipStream_t s, uint8_t* d_text, uint32_t data_off, uint32_t data_len,
                               const uint8_t* prev_text, const TextSummary* prev_sum, uint32_t first_skip, bool eof, bool bed6,
                               void* d_scratch, size_t max_lines, int32_t* d_start, int32_t* d_end, uint8_t* d_mapq,
                               uint8_t* d_strand, TextSummary* d_sum) {
    const size_t n_max = (size_t)kTextCarryMax + data_len + 32;  // the range is only known on the device: launch for the most
    const unsigned n_blocks = (unsigned)((n_max + kTextBlockBytes - 1) / kTextBlockBytes);
    unsigned long long* state = static_cast<unsigned long long*>(d_scratch);
    const unsigned zero_blocks = std::min(64u, (n_blocks + 1 + 255) / 256);
    hipLaunchKernelGGL(piece_setup_kernel, dim3(1 + zero_blocks), dim3(256), 0, s, d_text, data_off, data_len, prev_text, prev_sum,
                       first_skip, eof ? 1 : 0, d_sum, state, n_blocks + 1);
    hipLaunchKernelGGL(lines_rows_kernel, dim3(n_blocks), dim3(kT), 0, s, d_text, (size_t)0, state, n_blocks, max_lines,
                       bed6 ? 1 : 0, d_start, d_end, d_mapq, d_strand, d_sum, 1);
}

void textparse_launch(hipStream_t s, const uint8_t* d_text, size_t n, bool bed6, void* d_scratch, size_t max_lines,
                      int32_t* d_start, int32_t* d_end, uint8_t* d_mapq, uint8_t* d_strand, TextSummary* d_sum) {
    if (n == 0) return;
    const unsigned n_blocks = (unsigned)((n + kTextBlockBytes - 1) / kTextBlockBytes);
    unsigned long long* state = static_cast<unsigned long long*>(d_scratch);
    (void)hipMemsetAsync(state, 0, ((size_t)n_blocks + 1) * sizeof(unsigned long long), s);
    hipLaunchKernelGGL(lines_rows_kernel, dim3(n_blocks), dim3(kT), 0, s, d_text, n, state, n_blocks, max_lines, bed6 ? 1 : 0,
                       d_start, d_end, d_mapq, d_strand, d_sum, 0);
}

}  // namespace ftk
