// Device-side row parser of the streaming text decoder: see ftk_textparse.h.
//
// Four small kernels per piece of inflated text (~170 MB, ~6.5 M rows), all HBM-streaming:
//   nl_count   16 bytes per thread -> newlines per 4 KB block
//   nl_scan    exclusive scan of the block counts (one block), total line count
//   nl_pos     the same 16 bytes again -> line_start[k + 1] = offset after the k-th newline
//   parse_rows one thread per line: name span, three unsigned decimals, strand, line end -> columns;
//              lines that are anything else are counted (the host then parses the piece itself); lines
//              whose contig name differs from the previous line's are listed (contig runs)
// Algorithmic bytes: the text twice plus 10 B per row out - well under a millisecond per piece, i.e.
// nothing next to the host's inflate of the same piece; the point is to take the parse off the host cores.
#include <hip/hip_runtime.h>

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

__device__ __forceinline__ unsigned nl_mask16(const uint8_t* __restrict__ text, size_t off, size_t n) {
    unsigned m = 0;
    if (off + kB <= n) {
        const uint4 v = *reinterpret_cast<const uint4*>(text + off);
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (((w[k] >> (8 * j)) & 0xffu) == (unsigned)'\n') m |= 1u << (4 * k + j);
    } else {
        for (int j = 0; j < kB && off + j < n; ++j)
            if (text[off + j] == '\n') m |= 1u << j;
    }
    return m;
}

__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d, 64);
    return v;
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

__global__ __launch_bounds__(kT) void nl_count_kernel(const uint8_t* __restrict__ text_, size_t n_,
                                                      uint32_t* __restrict__ block_count, const TextSummary* ind) {
    __shared__ unsigned part[kT / 64];
    const Range R = range_of(text_, n_, ind);
    const uint8_t* __restrict__ text = R.text;
    const size_t n = R.n;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t off = ((size_t)blockIdx.x * kT + tid) * kB;
    unsigned m0 = off < n ? nl_mask16(text, off, n) : 0;
    if (off == 0) m0 &= ~0u << R.lead;
    unsigned c = __popc(m0);
    c = wave_sum(c);
    if (lane == 0) part[wv] = c;
    __syncthreads();
    if (tid == 0) {
        unsigned t = 0;
#pragma unroll
        for (int k = 0; k < kT / 64; ++k) t += part[k];
        block_count[blockIdx.x] = t;
    }
}

// exclusive scan of block_count[0, n_blocks) in place; block_count[n_blocks] = total = sum->n_lines
__global__ __launch_bounds__(1024) void nl_scan_kernel(uint32_t* __restrict__ block_count, int n_blocks_,
                                                       TextSummary* __restrict__ sum, int indirect) {
    __shared__ unsigned wave_tot[16];
    int n_blocks = n_blocks_;
    if (indirect) {
        const Range R = range_of(nullptr, 0, sum);
        n_blocks = (int)((R.n + kTextBlockBytes - 1) / kTextBlockBytes);
    }
    __shared__ unsigned base_of_wave[16];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int per = (n_blocks + 1023) / 1024;
    const int lo = tid * per, hi = min(lo + per, n_blocks);
    unsigned local = 0;
    for (int k = lo; k < hi; ++k) local += block_count[k];
    const unsigned incl = wave_incl_scan(local, lane);
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0;
        for (int k = 0; k < 16; ++k) { base_of_wave[k] = run; run += wave_tot[k]; }
        block_count[n_blocks] = run;
        sum->n_lines = run;
    }
    __syncthreads();
    unsigned run = base_of_wave[wv] + incl - local;
    for (int k = lo; k < hi; ++k) {
        const unsigned c = block_count[k];
        block_count[k] = run;
        run += c;
    }
}

__global__ __launch_bounds__(kT) void nl_pos_kernel(const uint8_t* __restrict__ text_, size_t n_,
                                                    const uint32_t* __restrict__ block_base,
                                                    uint32_t* __restrict__ line_start, size_t max_lines,
                                                    const TextSummary* ind) {
    __shared__ unsigned wave_tot[kT / 64];
    const Range R = range_of(text_, n_, ind);
    const uint8_t* __restrict__ text = R.text;
    const size_t n = R.n;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t off = ((size_t)blockIdx.x * kT + tid) * kB;
    unsigned m = off < n ? nl_mask16(text, off, n) : 0;
    if (off == 0) m &= ~0u << R.lead;
    const unsigned c = __popc(m);
    const unsigned incl = wave_incl_scan(c, lane);
    if (lane == 63) wave_tot[wv] = incl;
    __syncthreads();
    unsigned before = 0;
    for (int k = 0; k < wv; ++k) before += wave_tot[k];
    size_t idx = (size_t)block_base[blockIdx.x] + before + incl - c;  // newlines before this thread's bytes
    if (blockIdx.x == 0 && tid == 0) line_start[0] = R.lead;
    while (m) {
        const int j = __ffs(m) - 1;
        m &= m - 1;
        if (idx + 1 <= max_lines + 1) line_start[idx + 1] = (uint32_t)(off + j + 1);
        ++idx;
    }
}

// 1..10 decimal digits ending in `term` before the line end `e` (text[e] is the '\n'); p moves past `term`
__device__ __forceinline__ bool dev_digits(const uint8_t* __restrict__ t, uint32_t& p, uint32_t e, uint8_t term,
                                           unsigned long long& v) {
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

// The same row the host's one-pass parser accepts (plain_row in ftk_decode.cpp), with any contig name.
__global__ __launch_bounds__(256) void parse_rows_kernel(const uint8_t* __restrict__ t_,
                                                         const uint32_t* __restrict__ line_start, size_t max_lines,
                                                         int bed6, int32_t* __restrict__ o_start,
                                                         int32_t* __restrict__ o_end, uint8_t* __restrict__ o_mapq,
                                                         uint8_t* __restrict__ o_strand, TextSummary* __restrict__ sum,
                                                         int indirect) {
    const uint8_t* __restrict__ t = indirect ? range_of(t_, 0, sum).text : t_;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned long long n_lines = sum->n_lines;
    if (n_lines > max_lines) {
        if (i == 0) sum->overflow = 1;
        return;
    }
    bool bad = false;
    if (i < n_lines) {
        const uint32_t b = line_start[i], e = line_start[i + 1] - 1;  // text[e] == '\n'
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
        if (ok) {
            o_start[i] = (int32_t)fs;
            o_end[i] = (int32_t)fe;
            o_mapq[i] = (uint8_t)(mq < 255 ? mq : 255);
            o_strand[i] = strand == '+' ? 1 : 0;
            bool new_run = i == 0;
            if (!new_run) {
                const uint32_t pb = line_start[i - 1], pe = b - 1;
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
        bad = !ok;
        if (bad && i + 1 == n_lines) sum->last_line_bad = 1;
    }
    const unsigned long long mask = __ballot(bad);
    if ((threadIdx.x & 63) == 0 && mask) atomicAdd(&sum->n_bad, (unsigned long long)__popcll(mask));
}

// Device-inflated pieces: carry, skip, last line end (see textparse_launch_inflated).  One block.
__global__ __launch_bounds__(256) void piece_setup_kernel(uint8_t* __restrict__ text, uint32_t data_off, uint32_t data_len,
                                                          const uint8_t* __restrict__ prev_text,
                                                          const TextSummary* __restrict__ prev_sum, uint32_t first_skip,
                                                          int eof, TextSummary* __restrict__ sum) {
    __shared__ unsigned best;
    const int tid = threadIdx.x;
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

}  // namespace

void textparse_launch_inflated(hipStream_t s, uint8_t* d_text, uint32_t data_off, uint32_t data_len,
                               const uint8_t* prev_text, const TextSummary* prev_sum, uint32_t first_skip, bool eof, bool bed6,
                               uint32_t* d_block_count, uint32_t* d_line_start, size_t max_lines, int32_t* d_start,
                               int32_t* d_end, uint8_t* d_mapq, uint8_t* d_strand, TextSummary* d_sum) {
    hipLaunchKernelGGL(piece_setup_kernel, dim3(1), dim3(256), 0, s, d_text, data_off, data_len, prev_text, prev_sum,
                       first_skip, eof ? 1 : 0, d_sum);
    const size_t n_max = (size_t)kTextCarryMax + data_len + 32;  // the range is only known on the device: launch for the most
    const int n_blocks = (int)((n_max + kTextBlockBytes - 1) / kTextBlockBytes);
    hipLaunchKernelGGL(nl_count_kernel, dim3(n_blocks), dim3(kT), 0, s, d_text, (size_t)0, d_block_count, d_sum);
    hipLaunchKernelGGL(nl_scan_kernel, dim3(1), dim3(1024), 0, s, d_block_count, n_blocks, d_sum, 1);
    hipLaunchKernelGGL(nl_pos_kernel, dim3(n_blocks), dim3(kT), 0, s, d_text, (size_t)0, d_block_count, d_line_start, max_lines,
                       d_sum);
    const size_t row_blocks = (max_lines + 255) / 256;
    hipLaunchKernelGGL(parse_rows_kernel, dim3((unsigned)row_blocks), dim3(256), 0, s, d_text, d_line_start, max_lines,
                       bed6 ? 1 : 0, d_start, d_end, d_mapq, d_strand, d_sum, 1);
}

void textparse_launch(hipStream_t s, const uint8_t* d_text, size_t n, bool bed6, uint32_t* d_block_count,
                      uint32_t* d_line_start, size_t max_lines, int32_t* d_start, int32_t* d_end, uint8_t* d_mapq,
                      uint8_t* d_strand, TextSummary* d_sum) {
    if (n == 0) return;
    const int n_blocks = (int)((n + kTextBlockBytes - 1) / kTextBlockBytes);
    hipLaunchKernelGGL(nl_count_kernel, dim3(n_blocks), dim3(kT), 0, s, d_text, n, d_block_count, (const TextSummary*)nullptr);
    hipLaunchKernelGGL(nl_scan_kernel, dim3(1), dim3(1024), 0, s, d_block_count, n_blocks, d_sum, 0);
    hipLaunchKernelGGL(nl_pos_kernel, dim3(n_blocks), dim3(kT), 0, s, d_text, n, d_block_count, d_line_start, max_lines,
                       (const TextSummary*)nullptr);
    const size_t row_blocks = (max_lines + 255) / 256;
    hipLaunchKernelGGL(parse_rows_kernel, dim3((unsigned)row_blocks), dim3(256), 0, s, d_text, d_line_start, max_lines,
                       bed6 ? 1 : 0, d_start, d_end, d_mapq, d_strand, d_sum, 0);
}

}  // namespace ftk
