// BAM records -> fragment columns on gfx950 (see ftk_bamparse.h).
//
// Shape of the work.  A piece of a BAM stream is ~190 MB of inflated records, each starting where the previous one
// ends (block_size links): a chain of ~10^6 dependent loads if walked by one thread.  The range is cut into
// STRETCHES of 16 KB; one thread per stretch guesses where the first record of its stretch starts (header
// plausibility, two links deep - the host decoder's rule), walks and parses from there, and records where it landed.
// A stretch whose guess is not where the previous stretch landed is walked again from the true offset (a few
// Jacobi passes: every pass settles the stretches whose predecessor is settled); a last kernel checks that the
// whole chain is consistent, counts, scans the counts, and the emit pass writes the rows in file order.  All of it
// is memory-latency work - one dependent 36-byte header per record and thread - that the chip hides by running
// ~10^4 such chains side by side.  The results never depend on the guesses: an inconsistent chain is reported and
// the host walks that piece instead.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>

#include "ftk_bamparse.h"
#include "ftk_bamrule.h"

namespace ftk {
namespace {

typedef uint32_t __attribute__((aligned(1))) u32u;
typedef int32_t __attribute__((aligned(1))) i32u;
typedef uint16_t __attribute__((aligned(1))) u16u;

__device__ __forceinline__ uint32_t rd_u32(const uint8_t* p) { return *reinterpret_cast<const u32u*>(p); }
__device__ __forceinline__ int32_t rd_i32(const uint8_t* p) { return *reinterpret_cast<const i32u*>(p); }
__device__ __forceinline__ uint32_t rd_u16(const uint8_t* p) { return *reinterpret_cast<const u16u*>(p); }

// ftk_decode.cpp: plausible_record - does p[o..) look like the start of an alignment record?
__device__ bool plausible_record(const uint8_t* p, uint32_t o, uint32_t m, int n_ref) {
    if ((uint64_t)o + 36 > m) return false;
    const uint32_t bs = rd_u32(p + o);
    if (bs < 32 || bs > (1u << 24)) return false;
    const uint8_t* r = p + o + 4;
    const int32_t ref = rd_i32(r), pos = rd_i32(r + 4), next_ref = rd_i32(r + 20), next_pos = rd_i32(r + 24);
    if (ref < -1 || ref >= n_ref || next_ref < -1 || next_ref >= n_ref || pos < -1 || next_pos < -1) return false;
    const uint32_t l_name = r[8], n_cigar = rd_u16(r + 12);
    const int32_t l_seq = rd_i32(r + 16);
    if (l_name < 2 || l_seq < 0) return false;
    if (rd_u16(r + 14) & 0xf000u) return false;
    const uint64_t need = 32 + (uint64_t)l_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
    if (need > bs) return false;
    if ((uint64_t)o + 36 + l_name <= m) {
        if (r[32 + l_name - 1] != 0) return false;
        for (uint32_t k = 0; k + 1 < l_name; ++k)
            if (r[32 + k] < 33 || r[32 + k] > 126) return false;
    }
    return true;
}

// ftk_decode.cpp: guess_record_start
__device__ uint32_t guess_record_start(const uint8_t* p, uint32_t from, uint32_t m, int n_ref) {
    uint32_t fallback = kBamNoStart;
    for (uint64_t o64 = from; o64 + 36 <= m; ++o64) {
        const uint32_t o = (uint32_t)o64;
        if (!plausible_record(p, o, m, n_ref)) continue;
        uint64_t o2 = (uint64_t)o + 4 + rd_u32(p + o);
        bool ok = true;
        int checked = 0;
        for (int k = 0; k < 2; ++k) {
            if (o2 + 36 > m) break;
            if (!plausible_record(p, (uint32_t)o2, m, n_ref)) { ok = false; break; }
            ++checked;
            o2 += 4 + (uint64_t)rd_u32(p + o2);
        }
        if (!ok) continue;
        if (checked) return o;
        if (fallback == kBamNoStart) fallback = o;
        if (o - from > (1u << 16)) break;
    }
    return fallback;
}

struct RdDev {
    __device__ __forceinline__ uint32_t operator()(const uint8_t* p) const { return rd_u32(p); }
};

struct Out {
    int32_t *start, *end, *r1s, *r1e, *ref;
    uint8_t *mapq, *strand;
};

// One stretch: the records that START in [from, until), in order (run_bam's `walk`).  EMIT writes the fragments at
// row `row0`...; returns the landing offset, the fragment count and whether a bad block_size ended the walk.
template <bool EMIT>
__device__ uint32_t walk(const uint8_t* p, uint32_t m, uint32_t from, uint32_t until, const uint8_t* wanted, int n_ref,
                         uint32_t& n_frag, uint32_t& n_rec, bool& bad, const Out& out, uint32_t row0, size_t max_rows,
                         unsigned long long* last_key = nullptr, uint32_t* skipped = nullptr) {
    uint64_t o = from;
    n_frag = 0;
    n_rec = 0;
    bad = false;
    while (o < until && o + 4 <= m) {
        const uint32_t bs = rd_u32(p + o);
        if (bs < 32) { bad = true; break; }
        if (o + 4 + (uint64_t)bs > m) break;  // incomplete: waits for the next piece
        const uint8_t* r = p + o + 4;
        const int32_t ref_id = rd_i32(r);
        ++n_rec;
        if (EMIT && last_key) *last_key = ((unsigned long long)(uint32_t)ref_id << 32) | (uint32_t)max(rd_i32(r + 4), 0);
        if (ref_id >= 0 && ref_id < n_ref && wanted[ref_id]) {
            BamRow f;
            const int rule = bam_rule(r, bs, RdDev{}, f);  // io/alignment.py:60-71,242-268 (ftk_bamrule.h)
            if (rule == kBamFragment) {
                if (EMIT) {
                    const size_t i = (size_t)row0 + n_frag;
                    if (i < max_rows) {
                        out.start[i] = f.fs;
                        out.end[i] = f.fe;
                        out.mapq[i] = f.mapq;
                        out.strand[i] = f.fwd;
                        out.r1s[i] = f.r1s;
                        out.r1e[i] = f.r1e;
                        out.ref[i] = ref_id;
                    }
                }
                ++n_frag;
            } else if (EMIT && rule != kBamNotFragment && skipped) {
                ++skipped[rule == kBamNoCigarReverse];
            }
        }
        o += 4 + (uint64_t)bs;
    }
    return (uint32_t)o;
}

// The previous piece's unfinished record in front of this piece's data; the chain's range.
__global__ __launch_bounds__(256) void bam_setup_kernel(uint8_t* __restrict__ text, uint32_t data_off, uint32_t data_len,
                                                        const uint8_t* __restrict__ prev_text, const BamSummary* __restrict__ prev_sum,
                                                        uint32_t first_off, uint32_t stretch_bytes, BamSummary* __restrict__ sum) {
    uint32_t carry = 0;
    const uint8_t* src = nullptr;
    if (prev_sum) {
        carry = prev_sum->m - prev_sum->landing;
        src = prev_text + prev_sum->base_off + prev_sum->landing;
    }
    const bool overflow = carry > data_off;
    if (overflow) carry = 0;
    const uint32_t skip = first_off < data_len ? first_off : data_len;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < carry; i += gridDim.x * 256u) text[data_off - carry + i] = src[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sum->base_off = data_off - carry + (carry ? 0u : skip);  // (a carry and a skip never come together: the
        sum->m = carry + data_len - (carry ? 0u : skip);         //  skip belongs to the first piece of a read)
        sum->carry_overflow = overflow ? 1u : 0u;
        const uint32_t m = carry + data_len - (carry ? 0u : skip);
        sum->n_stretch = m ? (m + stretch_bytes - 1) / stretch_bytes : 1u;
    }
}

// pass 0: guess + walk every stretch.  pass >= 1 (Jacobi): a stretch that does not start where its predecessor landed
// is walked again from there - but only when that predecessor itself starts where ITS predecessor landed: a landing
// is as good as the start it was walked from, and following the landing of a stretch whose own guess was wrong would
// replace a good start by a bad one and push the damage one stretch further per pass.
__global__ __launch_bounds__(64) void bam_walk_kernel(const uint8_t* __restrict__ text, const BamSummary* __restrict__ sum,
                                                      const uint8_t* __restrict__ wanted, int n_ref, uint32_t stretch_bytes,
                                                      uint32_t* __restrict__ st, int pass) {
    const uint32_t n_stretch = sum->n_stretch, m = sum->m;
    const uint8_t* p = text + sum->base_off;
    uint32_t* st_start = st;
    uint32_t* st_land = st + (size_t)n_stretch;
    uint32_t* st_cnt = st + 2 * (size_t)n_stretch;
    for (uint32_t k = blockIdx.x * 64u + threadIdx.x; k < n_stretch; k += gridDim.x * 64u) {
        const uint64_t b0 = (uint64_t)k * stretch_bytes, b1 = b0 + stretch_bytes;
        const uint32_t until = b1 < m ? (uint32_t)b1 : m;
        uint32_t from;
        if (pass == 0) {
            from = k == 0 ? 0u : guess_record_start(p, (uint32_t)b0, m, n_ref);
            if (from == kBamNoStart) {
                st_start[k] = kBamNoStart;
                st_land[k] = kBamNoStart;
                st_cnt[k] = 0;
                continue;
            }
        } else {
            if (k == 0) continue;  // (walked from 0 in pass 0: final)
            from = st_land[k - 1];
            if (from == kBamNoStart || from == st_start[k]) continue;
            if (k >= 2 && st_start[k - 1] != st_land[k - 2]) continue;  // the predecessor is not settled itself
        }
        uint32_t n_frag, n_rec;
        bool bad;
        const uint32_t land = walk<false>(p, m, from, until, wanted, n_ref, n_frag, n_rec, bad, Out{}, 0, 0);
        st_start[k] = from;
        st_cnt[k] = n_frag | (bad ? 0x80000000u : 0u);
        st_land[k] = land;
    }
}

// one block: settle what the Jacobi passes left (serially, from the first stretch that does not start where its
// predecessor landed - every stretch before it is final), then: is the chain consistent, how many rows, where does
// every stretch's first row go
constexpr int kBamSerialRepairs = 4096;

__global__ __launch_bounds__(1024) void bam_scan_kernel(const uint8_t* __restrict__ text, BamSummary* __restrict__ sum,
                                                        const uint8_t* __restrict__ wanted, int n_ref, uint32_t stretch_bytes,
                                                        uint32_t* __restrict__ st) {
    __shared__ uint32_t part[1024];
    __shared__ uint32_t base_s, ok_s, bad_s, first_s;
    const uint32_t n_stretch = sum->n_stretch, m = sum->m;
    const uint8_t* p = text + sum->base_off;
    uint32_t* st_start = st;
    uint32_t* st_land = st + (size_t)n_stretch;
    uint32_t* st_cnt = st + 2 * (size_t)n_stretch;
    uint32_t* st_off = st + 3 * (size_t)n_stretch;
    const uint32_t tid = threadIdx.x;
    // ---- serial repair: the first unsettled stretch has a settled predecessor ---------------------------------
    uint32_t cursor = 1;
    int rep = 0;
    for (; rep < kBamSerialRepairs; ++rep) {
        if (tid == 0) first_s = 0xffffffffu;
        __syncthreads();
        uint32_t mine = 0xffffffffu;
        for (uint32_t k = cursor + tid; k < n_stretch; k += 1024)
            if (st_start[k] != st_land[k - 1]) { mine = k; break; }
        if (mine != 0xffffffffu) atomicMin(&first_s, mine);
        __syncthreads();
        const uint32_t f = first_s;
        if (f == 0xffffffffu) break;
        if (tid == 0) {
            const uint32_t from = st_land[f - 1];
            if (from != kBamNoStart) {
                const uint64_t b1 = ((uint64_t)f + 1) * stretch_bytes;
                const uint32_t until = b1 < m ? (uint32_t)b1 : m;
                uint32_t n_frag, n_rec;
                bool bad;
                const uint32_t land = walk<false>(p, m, from, until, wanted, n_ref, n_frag, n_rec, bad, Out{}, 0, 0);
                st_start[f] = from;
                st_cnt[f] = n_frag | (bad ? 0x80000000u : 0u);
                st_land[f] = land;
            }
            __threadfence_block();
        }
        __syncthreads();
        if (st_land[f - 1] == kBamNoStart) break;  // nothing to walk from: reported as inconsistent below
        cursor = f + 1;
    }
    __syncthreads();
    if (tid == 0) { base_s = 0; ok_s = 1; bad_s = 0; first_s = 0xffffffffu; sum->n_repairs = (uint32_t)rep; }
    __syncthreads();
    for (uint32_t c0 = 0; c0 < n_stretch; c0 += 1024) {
        const uint32_t k = c0 + tid;
        uint32_t cnt = 0;
        if (k < n_stretch) {
            const uint32_t want = k == 0 ? 0u : st_land[k - 1];
            if (st_start[k] != want || want == kBamNoStart) { atomicAnd(&ok_s, 0u); atomicMin(&first_s, k); }
            const uint32_t c = st_cnt[k];
            if (c & 0x80000000u) atomicOr(&bad_s, 1u);
            cnt = c & 0x7fffffffu;
        }
        part[tid] = cnt;
        __syncthreads();
        for (uint32_t d = 1; d < 1024; d <<= 1) {  // inclusive scan (Hillis-Steele; 10 rounds of a rare kernel)
            const uint32_t v = tid >= d ? part[tid - d] : 0u;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (k < n_stretch) st_off[k] = base_s + part[tid] - cnt;
        __syncthreads();
        if (tid == 0) base_s += part[1023];
        __syncthreads();
    }
    if (tid == 0) {
        sum->n_rows = base_s;
        sum->consistent = ok_s;
        sum->bad = bad_s;
        sum->landing = ok_s ? st_land[n_stretch - 1] : 0u;
        sum->first_unsettled = first_s;
        if (!ok_s && first_s < n_stretch) {
            const uint32_t f = first_s;
            sum->dbg[0] = st_start[f];
            sum->dbg[1] = f ? st_land[f - 1] : 0u;
            sum->dbg[2] = f ? st_start[f - 1] : 0u;
            sum->dbg[3] = f > 1 ? st_land[f - 2] : 0u;
        }
    }
}

__global__ __launch_bounds__(64) void bam_emit_kernel(const uint8_t* __restrict__ text, BamSummary* __restrict__ sum,
                                                      const uint8_t* __restrict__ wanted, int n_ref, uint32_t stretch_bytes,
                                                      const uint32_t* __restrict__ st, Out out, size_t max_rows) {
    if (!sum->consistent || sum->carry_overflow) return;
    const uint32_t n_stretch = sum->n_stretch, m = sum->m;
    const uint8_t* p = text + sum->base_off;
    const uint32_t* st_start = st;
    const uint32_t* st_off = st + 3 * (size_t)n_stretch;
    uint32_t recs = 0;
    uint32_t skipped[2] = {0, 0};  // fragments the columns cannot hold; CIGAR-less read1 with TLEN < 0 (ftk_bamrule.h)
    for (uint32_t k = blockIdx.x * 64u + threadIdx.x; k < n_stretch; k += gridDim.x * 64u) {
        const uint64_t b1 = ((uint64_t)k + 1) * stretch_bytes;
        const uint32_t until = b1 < m ? (uint32_t)b1 : m;
        uint32_t n_frag, n_rec;
        bool bad;
        unsigned long long key = 0;
        (void)walk<true>(p, m, st_start[k], until, wanted, n_ref, n_frag, n_rec, bad, out, st_off[k], max_rows, &key, skipped);
        recs += n_rec;
        if (n_rec) atomicMax(&sum->last_key, key);  // (records are sorted by reference and position: the last one's)
    }
    if (recs) atomicAdd(&sum->n_records, recs);
    if (skipped[0]) atomicAdd(&sum->n_unrepresentable, skipped[0]);
    if (skipped[1]) atomicAdd(&sum->n_nocigar_reverse, skipped[1]);
}

// contig runs of the rows: row i starts a run when its reference differs from row i - 1's
__global__ __launch_bounds__(256) void bam_runs_kernel(const int32_t* __restrict__ ref, BamSummary* __restrict__ sum, size_t max_rows) {
    if (!sum->consistent || sum->carry_overflow) return;
    const size_t n = sum->n_rows < max_rows ? sum->n_rows : max_rows;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        if (i == 0 || ref[i] != ref[i - 1]) {
            const uint32_t slot = atomicAdd(&sum->n_runs, 1u);
            if (slot < (uint32_t)kBamMaxRuns) {
                sum->run_row[slot] = (uint32_t)i;
                sum->run_ref[slot] = ref[i];
            }
        }
    }
}

__global__ __launch_bounds__(256) void iota_kernel(uint32_t* v, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) v[i] = (uint32_t)i;
}

__global__ __launch_bounds__(256) void bam_gather_kernel(size_t n, const uint32_t* __restrict__ perm, const int32_t* in_start,
                                                         const int32_t* in_end, const uint8_t* in_mapq, const uint8_t* in_strand,
                                                         const int32_t* in_r1s, const int32_t* in_r1e, int32_t* out_start,
                                                         int32_t* out_end, uint8_t* out_mapq, uint8_t* out_strand,
                                                         int32_t* out_r1s, int32_t* out_r1e, int32_t* out_ord) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const uint32_t j = perm[i];
        out_start[i] = in_start[j];
        out_end[i] = in_end[j];
        out_mapq[i] = in_mapq[j];
        out_strand[i] = in_strand[j];
        out_r1s[i] = in_r1s[j];
        out_r1e[i] = in_r1e[j];
        out_ord[i] = (int32_t)j;
    }
}

size_t align256(size_t b) { return (b + 255) / 256 * 256; }

}  // namespace

void bamparse_launch(hipStream_t s, uint8_t* d_text, uint32_t data_off, uint32_t data_len, const uint8_t* prev_text,
                     const BamSummary* prev_sum, uint32_t first_off, const uint8_t* d_wanted, int n_ref,
                     uint32_t stretch_bytes, uint32_t* d_stretch, size_t stretch_words, size_t max_rows, int32_t* d_start,
                     int32_t* d_end, uint8_t* d_mapq, uint8_t* d_strand, int32_t* d_r1s, int32_t* d_r1e, int32_t* d_ref,
                     BamSummary* d_sum) {
    (void)stretch_words;
    hipLaunchKernelGGL(bam_setup_kernel, dim3(64), dim3(256), 0, s, d_text, data_off, data_len, prev_text, prev_sum, first_off,
                       stretch_bytes, d_sum);
    // (the range is only known on the device: launch for the data plus a 1 MB carry, the kernels stride over more)
    const size_t est = ((size_t)data_len + (size_t(1) << 20)) / stretch_bytes + 1;
    const unsigned blocks = (unsigned)std::max<size_t>(1, std::min<size_t>((est + 63) / 64, 1u << 16));
    constexpr int kFixPasses = 3;
    for (int pass = 0; pass <= kFixPasses; ++pass)
        hipLaunchKernelGGL(bam_walk_kernel, dim3(blocks), dim3(64), 0, s, d_text, d_sum, d_wanted, n_ref, stretch_bytes, d_stretch, pass);
    hipLaunchKernelGGL(bam_scan_kernel, dim3(1), dim3(1024), 0, s, d_text, d_sum, d_wanted, n_ref, stretch_bytes, d_stretch);
    Out out{d_start, d_end, d_r1s, d_r1e, d_ref, d_mapq, d_strand};
    hipLaunchKernelGGL(bam_emit_kernel, dim3(blocks), dim3(64), 0, s, d_text, d_sum, d_wanted, n_ref, stretch_bytes, d_stretch, out,
                       max_rows);
    hipLaunchKernelGGL(bam_runs_kernel, dim3(1024), dim3(256), 0, s, d_ref, d_sum, max_rows);
}

size_t bam_sort_tmp_bytes(size_t n) {
    size_t tmp = 0;
    (void)rocprim::radix_sort_pairs(nullptr, tmp, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr,
                                    (uint32_t*)nullptr, n, 0, 31);
    return 3 * align256(n * 4) + align256(tmp) + 256;
}

int bam_sort_contig(hipStream_t s, size_t n, const int32_t* in_start, const int32_t* in_end, const uint8_t* in_mapq,
                    const uint8_t* in_strand, const int32_t* in_r1s, const int32_t* in_r1e, int32_t* out_start,
                    int32_t* out_end, uint8_t* out_mapq, uint8_t* out_strand, int32_t* out_r1s, int32_t* out_r1e,
                    int32_t* out_ord, void* d_tmp, size_t tmp_bytes) {
    if (n == 0) return 0;
    char* q = (char*)d_tmp;
    uint32_t* keys_out = (uint32_t*)q;
    uint32_t* vals_in = (uint32_t*)(q + align256(n * 4));
    uint32_t* vals_out = (uint32_t*)(q + 2 * align256(n * 4));
    void* rp = q + 3 * align256(n * 4);
    size_t rp_bytes = tmp_bytes - 3 * align256(n * 4);
    const unsigned blocks = (unsigned)std::min<size_t>((n + 255) / 256, 1u << 16);
    hipLaunchKernelGGL(iota_kernel, dim3(blocks), dim3(256), 0, s, vals_in, n);
    // starts are in [0, 2^31): 31 key bits; the sort is stable, so equal starts keep their file order
    hipError_t e = rocprim::radix_sort_pairs(rp, rp_bytes, (const uint32_t*)in_start, keys_out, (const uint32_t*)vals_in, vals_out, n,
                                             0, 31, s);
    if (e != hipSuccess) return -1;
    hipLaunchKernelGGL(bam_gather_kernel, dim3(blocks), dim3(256), 0, s, n, vals_out, in_start, in_end, in_mapq, in_strand, in_r1s,
                       in_r1e, out_start, out_end, out_mapq, out_strand, out_r1s, out_r1e, out_ord);
    return 0;
}

}  // namespace ftk
