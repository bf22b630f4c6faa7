// C-ABI implementation (include/ftk.h): contexts, HBM residency of fragments,
// and the host side of every feature call.  No CPU compute fallback exists.
#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <future>
#include <memory>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <type_traits>

#include <dlfcn.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <fcntl.h>
#include <unistd.h>

#include "ftk_host.h"
#include "ftk_inflate.h"
#include "ftk_kernels.h"

using namespace ftk;

namespace {

thread_local std::string g_err;  // for ctx-less failures

int fail(ftk_ctx* ctx, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_err = buf;
    return code;
}

#define HIPCHK(ctx, call)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess) {                                                                 \
            (void)hipGetLastError();                                                            \
            return fail(ctx, e_ == hipErrorOutOfMemory ? FTK_ERR_OOM : FTK_ERR_HIP, "%s: %s",   \
                        #call, hipGetErrorString(e_));                                          \
        }                                                                                       \
    } while (0)

bool is_device_ptr(const void* p) {
    if (!p) return false;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Bump allocator over the ctx scratch; the total is reserved up front so the
// base never moves within one API call.
struct Arena {
    char* base;
    size_t off = 0, cap;
    Arena(ftk_ctx* c) : base((char*)c->scratch), cap(c->scratch_bytes) {}
    template <class T>
    T* take(size_t n) {
        size_t bytes = align_up(n * sizeof(T));
        T* p = reinterpret_cast<T*>(base + off);
        off += bytes;
        return p;
    }
};

int reserve_scratch(ftk_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->scratch_bytes) return FTK_OK;
    // earlier work on the stream may still read the old scratch
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    // The block comes from, and goes back to, the library's cache of idle device blocks: a context that is closed and
    // opened again (one per file in a long-lived process, one per repetition in bench.py) then finds its scratch there.
    // A fresh hipMalloc of 0.5 GB was seen to take 0.3-0.4 s now and then - when the process had just freed tens of GB
    // (the allocation waits for the driver to finish with the memory it is handed) - in a call whose work is 3 ms.
    if (ctx->scratch) ftk_host::device_block_give(ctx->scratch, ctx->scratch_bytes, ctx->device);
    ctx->scratch = nullptr;
    ctx->scratch_bytes = 0;
    size_t got = 0;
    ctx->scratch = ftk_host::device_block_take(align_up(bytes + bytes / 4, 1 << 20), ctx->device, &got);
    if (!ctx->scratch) return fail(ctx, FTK_ERR_OOM, "cannot allocate %zu bytes of device scratch", bytes);
    ctx->scratch_bytes = got;
    return FTK_OK;
}

int get_contig(ftk_ctx* ctx, int contig_id, ContigData** out) {
    auto it = ctx->contigs.find(contig_id);
    if (it == ctx->contigs.end()) return fail(ctx, FTK_ERR_NO_CONTIG, "contig id %d is not loaded", contig_id);
    *out = &it->second;
    return FTK_OK;
}

void free_contig(ContigData& c) {
    if (c.base) (void)hipFree(c.base);
    if (c.r1) (void)hipFree(c.r1);
    if (c.order) (void)hipFree(c.order);
    if (c.bin_idx) (void)hipFree(c.bin_idx);
    c = ContigData{};
}

int check_filter(ftk_ctx* ctx, const ftk_filter* f, const ContigData& c, bool allow_fetch = false) {
    if (!f) return fail(ctx, FTK_ERR_INVALID, "filter is NULL");
    if (f->policy != FTK_POLICY_MIDPOINT && f->policy != FTK_POLICY_ANY && !(allow_fetch && f->policy == FTK_POLICY_FETCH))
        return fail(ctx, FTK_ERR_INVALID, "unknown intersect policy %d", f->policy);
    if (f->fetch_mode != FTK_FETCH_TABIX && f->fetch_mode != FTK_FETCH_BAM_READ1)
        return fail(ctx, FTK_ERR_INVALID, "unknown fetch mode %d", f->fetch_mode);
    if (f->fetch_mode == FTK_FETCH_BAM_READ1 && !c.v.r1_start)
        return fail(ctx, FTK_ERR_INVALID, "FTK_FETCH_BAM_READ1 needs ftk_frags_set_read1 first");
    return FTK_OK;
}

// longest fragment that can pass the filter
int eff_lmax(const ftk_filter* f, const ContigData& c) {
    int l = c.max_len;
    if (f && f->max_len >= 0) l = std::min(l, f->max_len);
    return std::max(l, 0);
}

// Stage an input array on the device if the caller gave a host pointer.
template <class T>
int stage_in(ftk_ctx* ctx, const T* src, size_t n, T* dev_buf, const T** out) {
    if (is_device_ptr(src)) {
        *out = src;
        return FTK_OK;
    }
    HIPCHK(ctx, hipMemcpyAsync(dev_buf, src, n * sizeof(T), hipMemcpyHostToDevice, ctx->stream));
    *out = dev_buf;
    return FTK_OK;
}

int upload_common(ftk_ctx* ctx, int contig_id, const int32_t* start, const int32_t* end, const uint8_t* mapq,
                  const uint8_t* strand, int64_t n, hipMemcpyKind kind) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    if (n < 0 || n > (int64_t)INT32_MAX - 1024) return fail(ctx, FTK_ERR_INVALID, "fragment count %lld out of range", (long long)n);
    if (n > 0 && (!start || !end || !mapq)) return fail(ctx, FTK_ERR_INVALID, "NULL fragment column");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    auto it = ctx->contigs.find(contig_id);
    if (it != ctx->contigs.end()) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        free_contig(it->second);
        ctx->contigs.erase(it);
    }
    ContigData c;
    c.n = n;
    // columns padded to a multiple of 4 fragments (+4) so 16-byte loads never
    // leave the allocation
    size_t n_pad = ((size_t)n + 3) / 4 * 4 + 4;
    size_t b_i32 = align_up(n_pad * 4), b_u8 = align_up(n_pad);
    size_t total = 2 * b_i32 + 2 * b_u8;
    HIPCHK(ctx, hipMalloc(&c.base, total));
    char* base = (char*)c.base;
    int32_t* d_start = (int32_t*)base;
    int32_t* d_end = (int32_t*)(base + b_i32);
    uint8_t* d_mapq = (uint8_t*)(base + 2 * b_i32);
    uint8_t* d_strand = (uint8_t*)(base + 2 * b_i32 + b_u8);
    hipStream_t s = ctx->stream;
    hipError_t e = hipMemsetAsync(c.base, 0, total, s);
    // padding fragments sit at 2^30, beyond every coordinate: no window test accepts them, so the
    // feature kernels process whole groups of four without bounds checks
    if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)(d_start + n), kPadCoord, n_pad - (size_t)n, s);
    if (e == hipSuccess) e = hipMemsetD32Async((hipDeviceptr_t)(d_end + n), kPadCoord, n_pad - (size_t)n, s);
    if (e == hipSuccess && n > 0) {
        e = hipMemcpyAsync(d_start, start, n * 4, kind, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_end, end, n * 4, kind, s);
        if (e == hipSuccess) e = hipMemcpyAsync(d_mapq, mapq, n, kind, s);
        if (e == hipSuccess && strand) e = hipMemcpyAsync(d_strand, strand, n, kind, s);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        free_contig(c);
        return fail(ctx, FTK_ERR_HIP, "fragment upload failed: %s", hipGetErrorString(e));
    }
    // validate + summarise on the device
    // (the summary lives in a device word block kept by the ctx: a hipMalloc / hipFree pair per contig made every
    // load wait for the whole device - an asynchronous copy-back of the previous contig's scores included)
    FragStats init{0, INT32_MIN, INT32_MAX, INT32_MIN, INT32_MAX};
    FragStats h_st = init;
    if (!ctx->d_stats) e = hipMalloc(&ctx->d_stats, sizeof(FragStats));
    FragStats* d_st = (FragStats*)ctx->d_stats;
    if (e == hipSuccess) e = hipMemcpyAsync(d_st, &init, sizeof(init), hipMemcpyHostToDevice, s);
    if (e == hipSuccess && n > 0) launch_stats(s, d_start, d_end, (int)n, d_st);
    if (e == hipSuccess) e = hipMemcpyAsync(&h_st, d_st, sizeof(h_st), hipMemcpyDeviceToHost, s);
    // the last start (the largest: the columns are sorted) for the position index below - on the ctx stream with the
    // summary: a synchronous hipMemcpy here waited for whatever transfer the device had in flight, 7-29 ms per contig
    // behind the previous contig's per-base results (the whole-genome run with WPS: 0.6 -> see DESIGN section 5)
    int32_t max_start = 0;
    if (e == hipSuccess && n > 0) e = hipMemcpyAsync(&max_start, d_start + (n - 1), 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        free_contig(c);
        return fail(ctx, FTK_ERR_HIP, "fragment validation failed: %s", hipGetErrorString(e));
    }
    if (n > 0) {
        if (h_st.unsorted) {
            free_contig(c);
            return fail(ctx, FTK_ERR_UNSORTED, "fragments of contig %d are not sorted by start", contig_id);
        }
        if (h_st.min_len < 0 || h_st.min_start < 0 || h_st.max_end >= (1 << 30)) {
            free_contig(c);
            return fail(ctx, FTK_ERR_INVALID,
                        "contig %d: coordinates must satisfy 0 <= start <= end < 2^30 (min len %d, min start %d, max end %d)",
                        contig_id, h_st.min_len, h_st.min_start, h_st.max_end);
        }
        c.max_len = h_st.max_len;
        c.max_end = h_st.max_end;
    }
    // coarse position index
    int n_bins = (max_start >> kBinShift) + 1;
    e = hipMalloc((void**)&c.bin_idx, (size_t)(n_bins + 1) * 4);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        free_contig(c);
        return fail(ctx, FTK_ERR_OOM, "index allocation failed");
    }
    launch_bin_index(s, d_start, (int)n, n_bins, c.bin_idx);
    e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        free_contig(c);
        return fail(ctx, FTK_ERR_HIP, "index build failed: %s", hipGetErrorString(e));
    }
    c.v.start = d_start;
    c.v.end = d_end;
    c.v.mapq = d_mapq;
    c.v.strand = d_strand;
    c.v.r1_start = nullptr;
    c.v.r1_end = nullptr;
    c.v.order = nullptr;
    c.v.bin_idx = c.bin_idx;
    c.v.n = (int32_t)n;
    c.v.n_bins = n_bins;
    c.v.max_len = c.max_len;
    ctx->contigs[contig_id] = c;
    return FTK_OK;
}

// Shared body of the window features: stage windows, plan candidate ranges.
struct WindowCall {
    const int32_t* d_ws = nullptr;
    const int32_t* d_we = nullptr;
    WindowPlan plan{};
};

size_t window_scratch_bytes(int64_t n_win) {
    return 2 * align_up(n_win * 4) + 3 * align_up(n_win * 4) + align_up((n_win + 1) * 4);
}

int window_prepare(ftk_ctx* ctx, ContigData* c, Arena& a, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                   int lmax, int small_max, WindowCall* wc, int64_t* const zero[4] = nullptr, bool plan = true) {
    int32_t* b_ws = a.take<int32_t>(n_win);
    int32_t* b_we = a.take<int32_t>(n_win);
    wc->plan.cand_lo = a.take<int32_t>(n_win);
    wc->plan.cand_hi = a.take<int32_t>(n_win);
    wc->plan.nchunks = a.take<uint32_t>(n_win);
    wc->plan.chunk_off = a.take<uint32_t>(n_win + 1);
    int rc;
    const size_t gap = (size_t)((const char*)b_we - (const char*)b_ws), bytes = gap + (size_t)n_win * 4;
    if (!is_device_ptr(w_start) && !is_device_ptr(w_end) && n_win > 0 && bytes <= (16u << 10)) {
        // both arrays in ONE copy (the arena lays b_we out behind b_ws): a small host-to-device copy costs the stream
        // ~4.5 us whatever its size, and a window call on a resident contig is tens of microseconds.  Only up to
        // 16 KB: the runtime takes a slower route for larger pageable copies (2 x 9.7 KB as one copy: +9 us).
        static thread_local std::vector<char> both;
        both.resize(bytes);
        memcpy(both.data(), w_start, (size_t)n_win * 4);
        memcpy(both.data() + gap, w_end, (size_t)n_win * 4);
        HIPCHK(ctx, hipMemcpyAsync(b_ws, both.data(), bytes, hipMemcpyHostToDevice, ctx->stream));  // pageable: staged before it returns
        wc->d_ws = b_ws;
        wc->d_we = b_we;
    } else {
        rc = stage_in(ctx, w_start, n_win, b_ws, &wc->d_ws);
        if (rc) return rc;
        rc = stage_in(ctx, w_end, n_win, b_we, &wc->d_we);
        if (rc) return rc;
    }
    if (plan) launch_plan(ctx->stream, c->v, wc->d_ws, wc->d_we, (int)n_win, lmax, small_max, wc->plan, zero);
    return FTK_OK;
}

// Many windows of similar length (a bin tiling): one block per window balances well and needs no plan.
// Host window arrays only (device arrays take the planned path: nothing is known about them here).
bool windows_suit_block_path(const ftk_ctx* ctx, const ContigData& c, int lmax, const int32_t* ws, const int32_t* we,
                             int64_t n_win) {
    static const int force = getenv("FTK_FEAT_BLOCK") ? atoi(getenv("FTK_FEAT_BLOCK")) : -1;
    if (force >= 0) return force != 0;
    if (n_win < ctx->n_cu) return false;
    long long total = 0, longest = 0;
    for (int64_t i = 0; i < n_win; ++i) {
        const long long len = (long long)we[i] - (long long)ws[i];
        if (len <= 0) continue;
        total += len;
        longest = std::max(longest, len);
    }
    // (measured on a chr2-sized contig at 30x: faster than the planned passes for tilings from 500 bp to
    // 100 kb windows -- 866 vs 948 us, 285 vs 325, 103 vs 152, 88 vs 147, 71 vs 94 -- so density is not a criterion)
    (void)c;
    (void)lmax;
    return total > 0 && longest * n_win <= 8 * total;
}

}  // namespace

extern "C" {

const char* ftk_version(void) { return "ftk-hip 0.1.0 (gfx950)"; }

int ftk_device_count(int* n_out) {
    if (!n_out) return fail(nullptr, FTK_ERR_INVALID, "n_out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        n = 0;
    }
    *n_out = n;
    return FTK_OK;
}

int ftk_ctx_create(int device_id, ftk_ctx** out) {
    if (!out) return fail(nullptr, FTK_ERR_INVALID, "out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return fail(nullptr, FTK_ERR_NO_DEVICE, "no HIP device available (%s); this library has no CPU fallback",
                    e == hipSuccess ? "0 devices" : hipGetErrorString(e));
    }
    if (device_id < 0 || device_id >= n)
        return fail(nullptr, FTK_ERR_NO_DEVICE, "device %d out of range (have %d); no CPU fallback", device_id, n);
    ftk_ctx* ctx = new (std::nothrow) ftk_ctx();
    if (!ctx) return fail(nullptr, FTK_ERR_OOM, "out of host memory");
    ctx->device = device_id;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device_id)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&ctx->ev_start)) != hipSuccess || (e = hipEventCreate(&ctx->ev_stop)) != hipSuccess) {
        (void)hipGetLastError();
        int rc = fail(nullptr, FTK_ERR_HIP, "context setup failed: %s", hipGetErrorString(e));
        delete ctx;
        return rc;
    }
    ctx->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->stream = ctx->own_stream;
    *out = ctx;
    return FTK_OK;
}

void ftk_ctx_destroy(ftk_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    for (auto& kv : ctx->contigs) free_contig(kv.second);
    for (auto& m : ctx->delfi_cache) (void)hipFree(m.base);
    for (int k = 0; k < 2; ++k)
        if (ctx->batch_dev[k]) (void)hipFree(ctx->batch_dev[k]);
    for (auto& kv : ctx->refs) {
        (void)hipFree(kv.second.d);
        if (kv.second.d_nblk) (void)hipFree(kv.second.d_nblk);
    }
    for (auto& b : ctx->ref_pool) (void)hipFree(b.first);
    for (int k = 0; k < 2; ++k) {
        if (ctx->ref_stage[k]) (void)hipHostFree(ctx->ref_stage[k]);
        if (ctx->ref_stage_done[k]) (void)hipEventDestroy(ctx->ref_stage_done[k]);
    }
    for (int k = 0; k < 2; ++k) {
        if (ctx->narrow_stage[k]) ftk_host_free(ctx->narrow_stage[k]);
        if (ctx->narrow_done[k]) (void)hipEventDestroy(ctx->narrow_done[k]);
    }
    if (ctx->param_stage) (void)hipHostFree(ctx->param_stage);
    if (ctx->scratch) ftk_host::device_block_give(ctx->scratch, ctx->scratch_bytes, ctx->device);
    if (ctx->d_stats) (void)hipFree(ctx->d_stats);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    for (int k = 0; k < 2; ++k) {
        if (ctx->abuf[k]) (void)hipFree(ctx->abuf[k]);
        if (ctx->a_kernel_done[k]) (void)hipEventDestroy(ctx->a_kernel_done[k]);
        if (ctx->a_copy_done[k]) (void)hipEventDestroy(ctx->a_copy_done[k]);
    }
    if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
    if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
    for (hipEvent_t e : ctx->ev_slots)
        if (e) (void)hipEventDestroy(e);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

const char* ftk_last_error(ftk_ctx* ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int ftk_ctx_set_stream(ftk_ctx* ctx, void* hip_stream) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    return FTK_OK;
}

int ftk_ctx_sync(ftk_ctx* ctx) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->copy_stream) {  // asynchronous host results (ftk_wps_async)
        HIPCHK(ctx, hipStreamSynchronize(ctx->copy_stream));
        ctx->a_pending[0] = ctx->a_pending[1] = false;
    }
    return FTK_OK;
}

int ftk_timer_start(ftk_ctx* ctx) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    HIPCHK(ctx, hipEventRecord(ctx->ev_start, ctx->stream));
    return FTK_OK;
}

int ftk_timer_stop(ftk_ctx* ctx, float* ms_out) {
    if (!ctx || !ms_out) return fail(ctx, FTK_ERR_INVALID, "NULL argument");
    HIPCHK(ctx, hipEventRecord(ctx->ev_stop, ctx->stream));
    HIPCHK(ctx, hipEventSynchronize(ctx->ev_stop));
    HIPCHK(ctx, hipEventElapsedTime(ms_out, ctx->ev_start, ctx->ev_stop));
    return FTK_OK;
}

int ftk_event_record(ftk_ctx* ctx, int slot) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    if (slot < 0 || slot >= FTK_MAX_EVENTS) return fail(ctx, FTK_ERR_INVALID, "event slot %d out of range", slot);
    if (ctx->ev_slots.empty()) ctx->ev_slots.assign(FTK_MAX_EVENTS, nullptr);
    if (!ctx->ev_slots[slot]) HIPCHK(ctx, hipEventCreate(&ctx->ev_slots[slot]));
    HIPCHK(ctx, hipEventRecord(ctx->ev_slots[slot], ctx->stream));
    return FTK_OK;
}

int ftk_event_elapsed_ms(ftk_ctx* ctx, int slot_a, int slot_b, float* ms_out) {
    if (!ctx || !ms_out) return fail(ctx, FTK_ERR_INVALID, "NULL argument");
    if (slot_a < 0 || slot_b < 0 || slot_a >= FTK_MAX_EVENTS || slot_b >= FTK_MAX_EVENTS || ctx->ev_slots.empty() ||
        !ctx->ev_slots[slot_a] || !ctx->ev_slots[slot_b])
        return fail(ctx, FTK_ERR_INVALID, "event slot not recorded");
    HIPCHK(ctx, hipEventSynchronize(ctx->ev_slots[slot_b]));
    HIPCHK(ctx, hipEventElapsedTime(ms_out, ctx->ev_slots[slot_a], ctx->ev_slots[slot_b]));
    return FTK_OK;
}

int ftk_frags_from_host(ftk_ctx* ctx, int contig_id, const int32_t* start, const int32_t* end, const uint8_t* mapq,
                        const uint8_t* strand, int64_t n) {
    return upload_common(ctx, contig_id, start, end, mapq, strand, n, hipMemcpyHostToDevice);
}

int ftk_frags_from_device(ftk_ctx* ctx, int contig_id, const int32_t* d_start, const int32_t* d_end,
                          const uint8_t* d_mapq, const uint8_t* d_strand, int64_t n) {
    return upload_common(ctx, contig_id, d_start, d_end, d_mapq, d_strand, n, hipMemcpyDeviceToDevice);
}

int ftk_frags_from_table(ftk_ctx* ctx, int contig_id, const ftk_fragtable* t, int i) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    const int32_t *s0 = nullptr, *e0 = nullptr, *r1s = nullptr, *r1e = nullptr;
    const uint8_t *q0 = nullptr, *st0 = nullptr;
    if (ftk_fragtable_columns(t, i, &s0, &e0, &q0, &st0, &r1s, &r1e) != FTK_OK)
        return fail(ctx, FTK_ERR_NO_CONTIG, "table has no contig %d", i);
    const int64_t n = ftk_fragtable_contig_rows(t, i);
    if (ftk_fragtable_is_device(t, i)) {
        // columns parsed on the GPU (ftk_fragstream_open_device): ordered behind the parse stream's last write
        HIPCHK(ctx, hipSetDevice(ctx->device));
        HIPCHK(ctx, hipStreamWaitEvent(ctx->stream, (hipEvent_t)ftk_fragtable_ready_event(t, i), 0));
        int rc = upload_common(ctx, contig_id, s0, e0, q0, st0, n, hipMemcpyDeviceToDevice);
        if (rc) return rc;
        // BAM records parsed on the device (run_bam_device): the read1 span and the file-order rank are device columns too
        if (r1s && r1e) rc = ftk_frags_set_read1(ctx, contig_id, r1s, r1e, n);
        const int32_t* dord = nullptr;
        if (rc == FTK_OK && r1s && ftk_fragtable_order(t, i, &dord) == FTK_OK && dord) rc = ftk_frags_set_order(ctx, contig_id, dord, n);
        return rc;
    }
    int rc = upload_common(ctx, contig_id, s0, e0, q0, st0, n, hipMemcpyHostToDevice);
    if (rc) return rc;
    if (r1s && r1e) rc = ftk_frags_set_read1(ctx, contig_id, r1s, r1e, n);
    const int32_t* ord = nullptr;
    if (rc == FTK_OK && ftk_fragtable_order(t, i, &ord) == FTK_OK && ord) rc = ftk_frags_set_order(ctx, contig_id, ord, n);
    return rc;
}

static int load_file(ftk_ctx* ctx, const char* path, const char* contig, int is_bam, int n_threads,
                     int first_contig_id, int* n_loaded_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    if (!path) return fail(ctx, FTK_ERR_INVALID, "path is NULL");
    if (n_loaded_out) *n_loaded_out = 0;
    ftk_fragstream* st = nullptr;
    int rc = is_bam ? ftk_fragstream_open(path, contig, is_bam, n_threads, 2, &st)
                    : ftk_fragstream_open_device(ctx->device, path, contig, 0, n_threads, 2, &st);
    if (rc != FTK_OK) return fail(ctx, rc, "%s", ftk_fragtable_error());
    int id = first_contig_id, n = 0;
    for (;;) {
        ftk_fragtable* t = nullptr;
        rc = ftk_fragstream_next(st, &t);
        if (rc != FTK_OK) { fail(ctx, rc, "%s", ftk_fragtable_error()); break; }
        if (!t) break;
        rc = ftk_frags_from_table(ctx, id, t, 0);
        if (rc == FTK_OK) ctx->names[id] = ftk_fragtable_contig_name(t, 0);
        ftk_fragtable_free(t);
        if (rc != FTK_OK) break;
        ++id;
        ++n;
    }
    ftk_fragstream_close(st);
    if (n_loaded_out) *n_loaded_out = n;
    return rc;
}

int ftk_frags_load_fraggz(ftk_ctx* ctx, const char* path, const char* contig, int n_threads, int first_contig_id,
                          int* n_loaded_out) {
    return load_file(ctx, path, contig, 0, n_threads, first_contig_id, n_loaded_out);
}

int ftk_frags_load_bam(ftk_ctx* ctx, const char* path, const char* contig, int n_threads, int first_contig_id,
                       int* n_loaded_out) {
    return load_file(ctx, path, contig, 1, n_threads, first_contig_id, n_loaded_out);
}

const char* ftk_frags_name(ftk_ctx* ctx, int contig_id) {
    if (!ctx) return nullptr;
    auto it = ctx->names.find(contig_id);
    return it == ctx->names.end() ? nullptr : it->second.c_str();
}

int ftk_frags_set_order(ftk_ctx* ctx, int contig_id, const int32_t* order, int64_t n) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    ContigData* c;
    int rc = get_contig(ctx, contig_id, &c);
    if (rc) return rc;
    if (n != c->n) return fail(ctx, FTK_ERR_INVALID, "order column has %lld rows, contig has %lld", (long long)n, (long long)c->n);
    if (n > 0 && !order) return fail(ctx, FTK_ERR_INVALID, "order is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (c->order) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipFree(c->order));
        c->order = nullptr;
        c->v.order = nullptr;
    }
    HIPCHK(ctx, hipMalloc((void**)&c->order, align_up((size_t)n * 4 + 16)));
    if (n > 0)
        HIPCHK(ctx, hipMemcpy(c->order, order, n * 4, is_device_ptr(order) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice));
    c->v.order = c->order;
    return FTK_OK;
}

int ftk_frags_set_read1(ftk_ctx* ctx, int contig_id, const int32_t* r1_start, const int32_t* r1_end, int64_t n) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    ContigData* c;
    int rc = get_contig(ctx, contig_id, &c);
    if (rc) return rc;
    if (n != c->n) return fail(ctx, FTK_ERR_INVALID, "read1 columns have %lld rows, contig has %lld", (long long)n, (long long)c->n);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    size_t n_pad = ((size_t)n + 3) / 4 * 4 + 4;
    size_t b = align_up(n_pad * 4);
    if (c->r1) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        HIPCHK(ctx, hipFree(c->r1));
        c->r1 = nullptr;
    }
    HIPCHK(ctx, hipMalloc((void**)&c->r1, 2 * b));
    HIPCHK(ctx, hipMemsetD32Async((hipDeviceptr_t)c->r1, kPadCoord, 2 * b / 4, ctx->stream));
    hipMemcpyKind k = is_device_ptr(r1_start) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    if (n > 0) {
        HIPCHK(ctx, hipMemcpyAsync(c->r1, r1_start, n * 4, k, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync((char*)c->r1 + b, r1_end, n * 4, k, ctx->stream));
    }
    // do all fragments hold their read1 span?  (ContigView::r1_inside: the kernels then read these columns only for
    // fragments that cross a window bound)
    int bad = 0;
    if (!ctx->d_stats) HIPCHK(ctx, hipMalloc(&ctx->d_stats, sizeof(FragStats)));
    HIPCHK(ctx, hipMemsetAsync(ctx->d_stats, 0, sizeof(int), ctx->stream));
    launch_r1_inside(ctx->stream, c->v.start, c->v.end, c->r1, (const int32_t*)((char*)c->r1 + b), (int)n, (int*)ctx->d_stats);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipMemcpyAsync(&bad, ctx->d_stats, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    c->v.r1_start = c->r1;
    c->v.r1_end = (int32_t*)((char*)c->r1 + b);
    static const bool r1_always = getenv("FTK_R1_ALWAYS") && atoi(getenv("FTK_R1_ALWAYS")) != 0;  // tests: both code paths
    c->v.r1_inside = (bad == 0 && !r1_always) ? 1 : 0;
    return FTK_OK;
}

int ftk_frags_info(ftk_ctx* ctx, int contig_id, int64_t* n_out, int32_t* max_len_out, int32_t* max_end_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    ContigData* c;
    int rc = get_contig(ctx, contig_id, &c);
    if (rc) return rc;
    if (n_out) *n_out = c->n;
    if (max_len_out) *max_len_out = c->max_len;
    if (max_end_out) *max_end_out = c->max_end;
    return FTK_OK;
}

int ftk_frags_release(ftk_ctx* ctx, int contig_id) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    auto it = ctx->contigs.find(contig_id);
    if (it == ctx->contigs.end()) return fail(ctx, FTK_ERR_NO_CONTIG, "contig id %d is not loaded", contig_id);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    free_contig(it->second);
    ctx->contigs.erase(it);
    ctx->names.erase(contig_id);
    for (size_t i = ctx->delfi_cache.size(); i-- > 0;)
        if (ctx->delfi_cache[i].contig_id == contig_id) {
            (void)hipFree(ctx->delfi_cache[i].base);
            ctx->delfi_cache.erase(ctx->delfi_cache.begin() + i);
        }
    return FTK_OK;
}

// ---- window features: one implementation behind four entry points ----------------
namespace {

constexpr size_t kDelfiCacheMax = 256;
constexpr int kBatchMaxItems = 64;

// Device copy of a small host descriptor array, re-uploaded only when its bytes change.
int upload_batch_descriptors(ftk_ctx* ctx, int slot, const void* host, size_t bytes, void** dev_out) {
    std::vector<unsigned char>& last = ctx->batch_host[slot];
    if (last.size() == bytes && bytes && memcmp(last.data(), host, bytes) == 0) {
        *dev_out = ctx->batch_dev[slot];
        return FTK_OK;
    }
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // a launch may still read the previous copy
    if (bytes > ctx->batch_cap[slot]) {
        if (ctx->batch_dev[slot]) (void)hipFree(ctx->batch_dev[slot]);
        ctx->batch_dev[slot] = nullptr;
        ctx->batch_cap[slot] = 0;
        HIPCHK(ctx, hipMalloc(&ctx->batch_dev[slot], align_up(bytes, 4096)));
        ctx->batch_cap[slot] = align_up(bytes, 4096);
    }
    HIPCHK(ctx, hipMemcpy(ctx->batch_dev[slot], host, bytes, hipMemcpyHostToDevice));
    last.assign((const unsigned char*)host, (const unsigned char*)host + bytes);
    *dev_out = ctx->batch_dev[slot];
    return FTK_OK;
}

// Device-resident windows + per-window blacklist CSR, cached by content.
int get_delfi_meta(ftk_ctx* ctx, int contig_id, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                   const int32_t* bl_start, const int32_t* bl_end, int64_t n_bl, DelfiMeta** out) {
    // two independent 64-bit digests of the content (FNV-1a over bytes, a multiply-rotate mix over bytes with
    // a different constant and position weighting): an entry is reused only when both agree, together with the
    // contig and both counts
    uint64_t key = 1469598103934665603ull, key2 = 0x9E3779B97F4A7C15ull;
    auto mix = [&key, &key2](const void* p, size_t n) {
        const unsigned char* b = (const unsigned char*)p;
        for (size_t i = 0; i < n; ++i) {
            key = (key ^ b[i]) * 1099511628211ull;
            key2 = (key2 + b[i] + 1) * 0xD6E8FEB86659FD93ull;
            key2 ^= key2 >> 29;
        }
    };
    mix(&contig_id, sizeof(contig_id));
    mix(&n_win, sizeof(n_win));
    mix(&n_bl, sizeof(n_bl));
    mix(w_start, n_win * 4);
    mix(w_end, n_win * 4);
    if (n_bl) { mix(bl_start, n_bl * 4); mix(bl_end, n_bl * 4); }
    for (auto& m : ctx->delfi_cache)
        if (m.key == key && m.key2 == key2 && m.contig_id == contig_id && m.n_win == n_win && m.n_bl == n_bl) {
            *out = &m;
            return FTK_OK;
        }
    // Blacklist regions fully inside each window (frag/_delfi.py:110-126):
    // region start >= w_start (bisect on the sorted starts) and stop <= w_end.
    // Per window keep (r0, running max of r1); a fragment is blacklisted iff
    // max{r1 : r0 <= fs} > fe, which equals "some region has r0 <= fs and
    // fe < r1" (frag/_delfi.py:455-462) also for overlapping regions.
    std::vector<int32_t> off(n_win + 1, 0), r0, pm;
    if (n_bl > 0) {
        for (int64_t i = 1; i < n_bl; ++i)
            if (bl_start[i] < bl_start[i - 1]) return fail(ctx, FTK_ERR_INVALID, "blacklist must be sorted by start");
        for (int64_t w = 0; w < n_win; ++w) {
            const int32_t* lo = std::lower_bound(bl_start, bl_start + n_bl, w_start[w]);
            int32_t run = INT32_MIN;
            for (int64_t j = lo - bl_start; j < n_bl && bl_start[j] < w_end[w]; ++j) {
                if (bl_end[j] <= w_end[w]) {
                    run = std::max(run, bl_end[j]);
                    r0.push_back(bl_start[j]);
                    pm.push_back(run);
                }
            }
            if (r0.size() > (size_t)INT32_MAX) return fail(ctx, FTK_ERR_INVALID, "blacklist expansion too large");
            off[w + 1] = (int32_t)r0.size();
        }
    }
    if (ctx->delfi_cache.size() >= kDelfiCacheMax) {  // drop the oldest entry
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->delfi_cache.front().base);
        ctx->delfi_cache.erase(ctx->delfi_cache.begin());
    }
    DelfiMeta m;
    m.key = key;
    m.key2 = key2;
    m.contig_id = contig_id;
    m.n_win = n_win;
    m.n_bl = n_bl;
    m.n_r = r0.size();
    const size_t b_w = align_up(n_win * 4), b_o = align_up((n_win + 1) * 4), b_r = align_up(std::max<size_t>(m.n_r, 1) * 4);
    HIPCHK(ctx, hipMalloc(&m.base, 2 * b_w + b_o + 2 * b_r));
    char* q = (char*)m.base;
    m.d_ws = (int32_t*)q;
    m.d_we = (int32_t*)(q + b_w);
    m.d_off = (int32_t*)(q + 2 * b_w);
    m.d_r0 = (int32_t*)(q + 2 * b_w + b_o);
    m.d_pm = (int32_t*)(q + 2 * b_w + b_o + b_r);
    // on the ctx stream, one wait for that stream at the end: a synchronous hipMemcpy waits for whatever transfer
    // the device has in flight - a previous contig's per-base results on their way to the host, tens of ms
    hipStream_t st = ctx->stream;
    hipError_t e = hipMemcpyAsync(m.d_ws, w_start, n_win * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(m.d_we, w_end, n_win * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipMemcpyAsync(m.d_off, off.data(), (n_win + 1) * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && m.n_r) e = hipMemcpyAsync(m.d_r0, r0.data(), m.n_r * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess && m.n_r) e = hipMemcpyAsync(m.d_pm, pm.data(), m.n_r * 4, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);  // (the sources are the caller's arrays and locals of this call)
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(m.base);
        return fail(ctx, FTK_ERR_HIP, "DELFI metadata upload failed: %s", hipGetErrorString(e));
    }
    ctx->delfi_cache.push_back(m);
    *out = &ctx->delfi_cache.back();
    return FTK_OK;
}

struct FeatCall {
    const ftk_filter* f = nullptr;  // coverage / histogram predicate
    int64_t* count_out = nullptr;
    uint32_t* hist_out = nullptr;
    int64_t* overflow_out = nullptr;
    int32_t len_lo = 0, n_bins = 0;
    bool delfi = false;
    int32_t mapq_min = 0;
    const int32_t* bl_start = nullptr;
    const int32_t* bl_end = nullptr;
    int64_t n_bl = 0;
    const ftk_gaps* gaps = nullptr;
    int64_t *short_out = nullptr, *long_out = nullptr, *nfrag_out = nullptr;
    const MotifParams* motif = nullptr;  // hist_out = k-mer histogram, overflow_out = error counts
    // per-window length statistics computed on the device from the histogram rows (which then need not leave it):
    // stats_out[w][7] = mean median stdev min max total n_short (launch_window_stats)
    double* stats_out = nullptr;
    int32_t short_cut = 0;
    bool want_hist() const { return hist_out || stats_out; }
};

// tail: a whole-interval WPS to run in the same launch (see launch_window_features).  scratch_prefix > 0: the caller keeps
// that many bytes at the start of the ctx scratch for itself - the tail's scores when they are bound for host memory -
// and tail->out is taken to be the scratch base.
int features_common(ftk_ctx* ctx, int contig_id, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                    const FeatCall& fc, const WpsTail* tail = nullptr, bool* tail_merged = nullptr, size_t scratch_prefix = 0) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    ContigData* c;
    int rc = get_contig(ctx, contig_id, &c);
    if (rc) return rc;
    const bool ch = fc.count_out || fc.want_hist();
    if (ch && (rc = check_filter(ctx, fc.f, *c))) return rc;
    if (n_win < 0 || n_win > (1 << 30)) return fail(ctx, FTK_ERR_INVALID, "n_win out of range");
    if (fc.want_hist() && (fc.n_bins <= 0 || fc.n_bins > kHistMaxBins))
        return fail(ctx, FTK_ERR_INVALID, "n_bins must be in [1, %d]; split the length range", kHistMaxBins);
    if (n_win == 0) return FTK_OK;
    if (!w_start || !w_end) return fail(ctx, FTK_ERR_INVALID, "NULL window pointer");
    if (fc.hist_out && !fc.overflow_out) return fail(ctx, FTK_ERR_INVALID, "overflow_out is NULL");
    if (!ch && !fc.delfi) return fail(ctx, FTK_ERR_INVALID, "no feature requested");
    ftk_gaps g{};
    DelfiMeta* meta = nullptr;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (fc.delfi) {
        if (!fc.short_out || !fc.long_out) return fail(ctx, FTK_ERR_INVALID, "NULL DELFI output pointer");
        if (fc.n_bl < 0 || (fc.n_bl > 0 && (!fc.bl_start || !fc.bl_end)))
            return fail(ctx, FTK_ERR_INVALID, "bad blacklist arguments");
        if (fc.gaps) g = *fc.gaps;
        if (g.has_gaps && (g.n_telo < 0 || g.n_telo > FTK_MAX_TELOMERES))
            return fail(ctx, FTK_ERR_INVALID, "at most %d telomere intervals per contig are supported", FTK_MAX_TELOMERES);
        if (is_device_ptr(w_start) || is_device_ptr(w_end) || is_device_ptr(fc.bl_start))
            return fail(ctx, FTK_ERR_INVALID, "DELFI takes host window and blacklist arrays");
        if ((rc = get_delfi_meta(ctx, contig_id, w_start, w_end, n_win, fc.bl_start, fc.bl_end, fc.n_bl, &meta)))
            return rc;
    }
    const bool c_dev = is_device_ptr(fc.count_out), h_dev = is_device_ptr(fc.hist_out),
               o_dev = is_device_ptr(fc.overflow_out), s_dev = is_device_ptr(fc.short_out),
               l_dev = is_device_ptr(fc.long_out), n_dev = is_device_ptr(fc.nfrag_out);
    const bool st_dev = is_device_ptr(fc.stats_out);
    const size_t hist_elems = fc.want_hist() ? (size_t)n_win * (size_t)fc.n_bins : 0;
    size_t need = window_scratch_bytes(n_win) + 5 * align_up(n_win * 8) + (h_dev ? 0 : align_up(hist_elems * 4)) +
                  (fc.stats_out && !st_dev ? align_up((size_t)n_win * 7 * 8) : 0);
    if ((rc = reserve_scratch(ctx, scratch_prefix + need))) return rc;
    Arena a(ctx);
    a.off = scratch_prefix;
    WpsTail tail_here;
    if (tail && scratch_prefix) {
        tail_here = *tail;
        tail_here.out = (int64_t*)ctx->scratch;
        tail = &tail_here;
    }
    FeatureRequest r;
    r.filter = fc.f;
    r.cov_out = fc.count_out ? (c_dev ? fc.count_out : a.take<int64_t>(n_win)) : nullptr;
    r.hist_out = fc.want_hist() ? (h_dev ? fc.hist_out : a.take<uint32_t>(hist_elems)) : nullptr;
    r.over_out = fc.want_hist() ? (o_dev && fc.hist_out ? fc.overflow_out : a.take<int64_t>(n_win)) : nullptr;
    double* d_stats = fc.stats_out ? (st_dev ? fc.stats_out : a.take<double>((size_t)n_win * 7)) : nullptr;
    r.len_lo = fc.len_lo;
    r.n_bins = fc.n_bins;
    r.motif = fc.motif;
    int64_t* d_nfrag = nullptr;
    if (fc.delfi) {
        r.short_out = s_dev ? fc.short_out : a.take<int64_t>(n_win);
        r.long_out = l_dev ? fc.long_out : a.take<int64_t>(n_win);
        d_nfrag = fc.nfrag_out ? (n_dev ? fc.nfrag_out : a.take<int64_t>(n_win)) : nullptr;
        r.delfi_mapq_min = fc.mapq_min;
        r.gaps = g;
        if (meta->n_r) { r.bl_off = meta->d_off; r.bl_r0 = meta->d_r0; r.bl_pm = meta->d_pm; }
    }
    // longest fragment any requested feature can accept
    int lmax = 0;
    if (ch) lmax = std::max(lmax, eff_lmax(fc.f, *c));
    if (fc.delfi) lmax = std::max(lmax, std::max(0, std::min(220, c->max_len)));
    const bool small_path = !(fc.want_hist() && fc.n_bins > kHistSmallMaxBins);
    const bool block_path = !is_device_ptr(w_start) && !is_device_ptr(w_end) &&
                            windows_suit_block_path(ctx, *c, lmax, w_start, w_end, n_win);
    WindowCall wc;
    int64_t* zero[4] = {r.cov_out, r.short_out, r.long_out, nullptr};
    if ((rc = window_prepare(ctx, c, a, meta ? meta->d_ws : w_start, meta ? meta->d_we : w_end, n_win, lmax,
                             small_path ? kSmallMax : -1, &wc, zero, !block_path)))
        return rc;
    if (r.hist_out && !small_path && !block_path) {  // with the wave-per-window pass on, it writes / clears every row itself
        HIPCHK(ctx, hipMemsetAsync(r.hist_out, 0, hist_elems * 4, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(r.over_out, 0, n_win * 8, ctx->stream));
    }
    // blocks per CU of the chunk walker: 8 are resident, 32 give shorter per-block chunk ranges and a
    // smoother tail (measured best of 2..256 on the whole-genome bench); FTK_FEAT_BPC for experiments
    static const int bpc = getenv("FTK_FEAT_BPC") ? atoi(getenv("FTK_FEAT_BPC")) : 32;
    if (block_path && c->max_end > 0) {  // expected candidates per window from the contig's mean density
        double span = 0;
        for (int64_t i = 0; i < n_win; ++i) span += std::max(0.0, (double)w_end[i] - (double)w_start[i]);
        const double est = (double)c->n / (double)c->max_end * (span / (double)n_win + lmax);
        r.block_threads = est >= 4096.0 ? 512 : 256;
    }
    if (tail) r.block_threads = 256;
    const bool merged = launch_window_features(ctx->stream, ctx->n_cu * bpc, c->v, wc.d_ws, wc.d_we, (int)n_win, wc.plan,
                                               r, small_path, block_path ? lmax : -1, tail);
    if (tail_merged) *tail_merged = merged;
    if (d_nfrag) launch_add_i64(ctx->stream, r.short_out, r.long_out, d_nfrag, (int)n_win);
    if (d_stats) launch_window_stats(ctx->stream, r.hist_out, (int)n_win, fc.n_bins, fc.len_lo, fc.short_cut, d_stats);
    HIPCHK(ctx, hipGetLastError());
    bool host_out = false;
    auto back = [&](void* dst, const void* src, size_t bytes, bool dev) -> hipError_t {
        if (!dst || dev) return hipSuccess;
        host_out = true;
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    };
    HIPCHK(ctx, back(fc.count_out, r.cov_out, n_win * 8, c_dev));
    HIPCHK(ctx, back(fc.hist_out, r.hist_out, hist_elems * 4, h_dev));
    HIPCHK(ctx, back(fc.hist_out ? fc.overflow_out : nullptr, r.over_out, n_win * 8, o_dev));
    HIPCHK(ctx, back(fc.short_out, r.short_out, n_win * 8, s_dev));
    HIPCHK(ctx, back(fc.long_out, r.long_out, n_win * 8, l_dev));
    HIPCHK(ctx, back(fc.nfrag_out, d_nfrag, n_win * 8, n_dev));
    HIPCHK(ctx, back(fc.stats_out, d_stats, (size_t)n_win * 7 * 8, st_dev));
    if (host_out) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FTK_OK;
}

}  // namespace

int ftk_window_features_batch(ftk_ctx* ctx, const ftk_feature_item* items, int32_t n_items, const ftk_filter* f,
                              int64_t* count_out, int32_t len_lo, int32_t n_bins, uint32_t* hist_out,
                              int64_t* overflow_out, int32_t delfi_mapq_min, int64_t* short_out, int64_t* long_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    if (n_items < 0 || n_items > kBatchMaxItems)
        return fail(ctx, FTK_ERR_INVALID, "n_items must be in [0, %d]", kBatchMaxItems);
    if (n_items == 0) return FTK_OK;
    if (!items) return fail(ctx, FTK_ERR_INVALID, "items is NULL");
    const bool ch = count_out || hist_out, df = short_out || long_out;
    if (!ch && !df) return fail(ctx, FTK_ERR_INVALID, "no feature requested");
    if (df && (!short_out || !long_out)) return fail(ctx, FTK_ERR_INVALID, "NULL DELFI output pointer");
    if (hist_out && (n_bins <= 0 || n_bins > kHistMaxBins || !overflow_out))
        return fail(ctx, FTK_ERR_INVALID, "histogram needs n_bins in [1, %d] and overflow_out", kHistMaxBins);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // make room first: an eviction in the middle of the loop below could free an earlier item's arrays
    while (ctx->delfi_cache.size() + (size_t)n_items > kDelfiCacheMax) {
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->delfi_cache.front().base);
        ctx->delfi_cache.erase(ctx->delfi_cache.begin());
    }
    std::vector<FeatItem> host(n_items);
    int64_t total = 0;
    int n_bam = 0;
    for (int i = 0; i < n_items; ++i) {
        const ftk_feature_item& it = items[i];
        ContigData* c;
        int rc = get_contig(ctx, it.contig_id, &c);
        if (rc) return rc;
        if (ch && (rc = check_filter(ctx, f, *c))) return rc;
        if (it.n_win <= 0 || it.n_win > (1 << 30) || !it.w_start || !it.w_end)
            return fail(ctx, FTK_ERR_INVALID, "item %d: bad window arguments", i);
        if (it.n_bl < 0 || (it.n_bl > 0 && (!it.bl_start || !it.bl_end)))
            return fail(ctx, FTK_ERR_INVALID, "item %d: bad blacklist arguments", i);
        if (is_device_ptr(it.w_start) || is_device_ptr(it.w_end) || is_device_ptr(it.bl_start))
            return fail(ctx, FTK_ERR_INVALID, "item %d: windows and blacklist must be host arrays", i);
        ftk_gaps g{};
        if (it.gaps) g = *it.gaps;
        if (g.has_gaps && (g.n_telo < 0 || g.n_telo > FTK_MAX_TELOMERES))
            return fail(ctx, FTK_ERR_INVALID, "item %d: at most %d telomere intervals", i, FTK_MAX_TELOMERES);
        DelfiMeta* meta = nullptr;
        if ((rc = get_delfi_meta(ctx, it.contig_id, it.w_start, it.w_end, it.n_win, df ? it.bl_start : nullptr,
                                 df ? it.bl_end : nullptr, df ? it.n_bl : 0, &meta)))
            return rc;
        FeatItem& h = host[i];
        memset(&h, 0, sizeof(h));
        h.cv = c->v;
        h.ws = meta->d_ws;
        h.we = meta->d_we;
        if (df && meta->n_r) { h.bl_off = meta->d_off; h.bl_r0 = meta->d_r0; h.bl_pm = meta->d_pm; }
        h.win_base = (int32_t)total;
        h.n_win = (int32_t)it.n_win;
        int lmax = 0;
        if (ch) lmax = std::max(lmax, eff_lmax(f, *c));
        if (df) lmax = std::max(lmax, std::max(0, std::min(220, c->max_len)));
        h.lmax = lmax;
        int gc[4];
        gap_constants(g, gc);
        h.cen0 = gc[0]; h.cen1 = gc[1]; h.tel0 = gc[2]; h.tel1 = gc[3];
        total += it.n_win;
        n_bam += c->v.r1_start != nullptr && (!f || f->fetch_mode == FTK_FETCH_BAM_READ1);
        if (total > (1 << 30)) return fail(ctx, FTK_ERR_INVALID, "too many windows in one batch");
    }
    if (n_bam != 0 && n_bam != n_items)
        return fail(ctx, FTK_ERR_INVALID, "a batch cannot mix contigs with and without read1 columns");
    void* d_items = nullptr;
    int rc = upload_batch_descriptors(ctx, 0, host.data(), host.size() * sizeof(FeatItem), &d_items);
    if (rc) return rc;
    const bool c_dev = is_device_ptr(count_out), h_dev = is_device_ptr(hist_out), o_dev = is_device_ptr(overflow_out),
               s_dev = is_device_ptr(short_out), l_dev = is_device_ptr(long_out);
    const size_t hist_elems = hist_out ? (size_t)total * (size_t)n_bins : 0;
    if ((rc = reserve_scratch(ctx, 4 * align_up(total * 8) + (h_dev ? 0 : align_up(hist_elems * 4))))) return rc;
    Arena a(ctx);
    FeatureRequest r;
    r.filter = f;
    r.cov_out = count_out ? (c_dev ? count_out : a.take<int64_t>(total)) : nullptr;
    r.hist_out = hist_out ? (h_dev ? hist_out : a.take<uint32_t>(hist_elems)) : nullptr;
    r.over_out = hist_out ? (o_dev ? overflow_out : a.take<int64_t>(total)) : nullptr;
    r.len_lo = len_lo;
    r.n_bins = n_bins;
    if (df) {
        r.short_out = s_dev ? short_out : a.take<int64_t>(total);
        r.long_out = l_dev ? long_out : a.take<int64_t>(total);
        r.delfi_mapq_min = delfi_mapq_min;
    }
    launch_window_features_batch(ctx->stream, (const FeatItem*)d_items, n_items, (int)total, r, n_bam != 0);
    HIPCHK(ctx, hipGetLastError());
    bool host_out = false;
    auto back = [&](void* dst, const void* src, size_t bytes, bool dev) -> hipError_t {
        if (!dst || dev) return hipSuccess;
        host_out = true;
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    };
    HIPCHK(ctx, back(count_out, r.cov_out, total * 8, c_dev));
    HIPCHK(ctx, back(hist_out, r.hist_out, hist_elems * 4, h_dev));
    HIPCHK(ctx, back(hist_out ? overflow_out : nullptr, r.over_out, total * 8, o_dev));
    HIPCHK(ctx, back(short_out, r.short_out, total * 8, s_dev));
    HIPCHK(ctx, back(long_out, r.long_out, total * 8, l_dev));
    if (host_out) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FTK_OK;
}

int ftk_window_counts(ftk_ctx* ctx, int contig_id, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                      const ftk_filter* f, int64_t* count_out) {
    if (ctx && n_win > 0 && !count_out) return fail(ctx, FTK_ERR_INVALID, "count_out is NULL");
    FeatCall fc;
    fc.f = f;
    fc.count_out = count_out;
    return features_common(ctx, contig_id, w_start, w_end, n_win, fc);
}

int ftk_delfi_counts(ftk_ctx* ctx, int contig_id, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                     int32_t mapq_min, const int32_t* bl_start, const int32_t* bl_end, int64_t n_bl,
                     const ftk_gaps* gaps, int64_t* short_out, int64_t* long_out, int64_t* nfrag_out) {
    FeatCall fc;
    fc.delfi = true;
    fc.mapq_min = mapq_min;
    fc.bl_start = bl_start;
    fc.bl_end = bl_end;
    fc.n_bl = n_bl;
    fc.gaps = gaps;
    fc.short_out = short_out;
    fc.long_out = long_out;
    fc.nfrag_out = nfrag_out;
    return features_common(ctx, contig_id, w_start, w_end, n_win, fc);
}

int ftk_fraglen_hist(ftk_ctx* ctx, int contig_id, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                     const ftk_filter* f, int32_t len_lo, int32_t n_bins, uint32_t* hist_out, int64_t* overflow_out) {
    if (ctx && n_win > 0 && !hist_out) return fail(ctx, FTK_ERR_INVALID, "hist_out is NULL");
    FeatCall fc;
    fc.f = f;
    fc.hist_out = hist_out;
    fc.overflow_out = overflow_out;
    fc.len_lo = len_lo;
    fc.n_bins = n_bins;
    return features_common(ctx, contig_id, w_start, w_end, n_win, fc);
}

int ftk_fraglen_stats(ftk_ctx* ctx, int contig_id, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                      const ftk_filter* f, int32_t len_lo, int32_t n_bins, int32_t short_cut, double* stats_out) {
    if (ctx && n_win > 0 && !stats_out) return fail(ctx, FTK_ERR_INVALID, "stats_out is NULL");
    FeatCall fc;
    fc.f = f;
    fc.len_lo = len_lo;
    fc.n_bins = n_bins;
    fc.stats_out = stats_out;
    fc.short_cut = short_cut;
    return features_common(ctx, contig_id, w_start, w_end, n_win, fc);
}

int ftk_window_features(ftk_ctx* ctx, int contig_id, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                        const ftk_filter* f, int64_t* count_out, int32_t len_lo, int32_t n_bins, uint32_t* hist_out,
                        int64_t* overflow_out, int32_t delfi_mapq_min, const int32_t* bl_start, const int32_t* bl_end,
                        int64_t n_bl, const ftk_gaps* gaps, int64_t* short_out, int64_t* long_out) {
    FeatCall fc;
    fc.f = f;
    fc.count_out = count_out;
    fc.hist_out = hist_out;
    fc.overflow_out = overflow_out;
    fc.len_lo = len_lo;
    fc.n_bins = n_bins;
    fc.delfi = short_out != nullptr || long_out != nullptr;
    fc.mapq_min = delfi_mapq_min;
    fc.bl_start = bl_start;
    fc.bl_end = bl_end;
    fc.n_bl = n_bl;
    fc.gaps = gaps;
    fc.short_out = short_out;
    fc.long_out = long_out;
    return features_common(ctx, contig_id, w_start, w_end, n_win, fc);
}

static int select_common(ftk_ctx* ctx, int contig_id, int32_t w_start, int32_t w_end, const ftk_filter* f,
                         int32_t* len_out, int32_t* start_out, int32_t* end_out, uint8_t* mapq_out,
                         uint8_t* strand_out, int64_t cap, int64_t* n_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    ContigData* c;
    int rc = get_contig(ctx, contig_id, &c);
    if (rc) return rc;
    if ((rc = check_filter(ctx, f, *c, true))) return rc;
    if (!n_out || cap < 0) return fail(ctx, FTK_ERR_INVALID, "bad output arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // candidate range of the single window (planned on the device, read back)
    if ((rc = reserve_scratch(ctx, window_scratch_bytes(1)))) return rc;
    int32_t lohi[2];
    {
        Arena a(ctx);
        WindowCall wc;
        if ((rc = window_prepare(ctx, c, a, &w_start, &w_end, 1, eff_lmax(f, *c), kSmallMax, &wc))) return rc;
        HIPCHK(ctx, hipMemcpyAsync(&lohi[0], wc.plan.cand_lo, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipMemcpyAsync(&lohi[1], wc.plan.cand_hi, 4, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    const int lo = lohi[0], hi = lohi[1];
    const int64_t n_cand = hi - lo;
    if (n_cand <= 0) {
        *n_out = 0;
        return FTK_OK;
    }
    const int nb = (int)((n_cand + 255) / 256);
    const int64_t ncap = std::min<int64_t>(cap, n_cand);
    size_t need = 2 * align_up((size_t)(nb + 1) * 4) + 4 * align_up(ncap * 4) + 2 * align_up(ncap);
    if ((rc = reserve_scratch(ctx, need))) return rc;
    Arena a(ctx);
    uint32_t* d_cnt = a.take<uint32_t>(nb + 1);
    uint32_t* d_off = a.take<uint32_t>(nb + 1);
    int32_t* d_len = len_out ? a.take<int32_t>(ncap) : nullptr;
    int32_t* d_s = start_out ? a.take<int32_t>(ncap) : nullptr;
    int32_t* d_e = end_out ? a.take<int32_t>(ncap) : nullptr;
    uint8_t* d_q = mapq_out ? a.take<uint8_t>(ncap) : nullptr;
    uint8_t* d_st = strand_out ? a.take<uint8_t>(ncap) : nullptr;
    int32_t* d_ord = c->v.order ? a.take<int32_t>(ncap) : nullptr;
    launch_select_count(ctx->stream, c->v, lo, hi, w_start, w_end, *f, d_cnt);
    launch_scan_u32(ctx->stream, d_cnt, nb, d_off);
    launch_select_write(ctx->stream, c->v, lo, hi, w_start, w_end, *f, d_off, ncap, d_len, d_s, d_e, d_q, d_st, d_ord);
    HIPCHK(ctx, hipGetLastError());
    uint32_t total = 0;
    HIPCHK(ctx, hipMemcpyAsync(&total, d_off + nb, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    *n_out = total;
    const int64_t n_copy = std::min<int64_t>(total, ncap);
    if (n_copy > 0) {
        if (len_out) HIPCHK(ctx, hipMemcpyAsync(len_out, d_len, n_copy * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (start_out) HIPCHK(ctx, hipMemcpyAsync(start_out, d_s, n_copy * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (end_out) HIPCHK(ctx, hipMemcpyAsync(end_out, d_e, n_copy * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (mapq_out) HIPCHK(ctx, hipMemcpyAsync(mapq_out, d_q, n_copy, hipMemcpyDeviceToHost, ctx->stream));
        if (strand_out) HIPCHK(ctx, hipMemcpyAsync(strand_out, d_st, n_copy, hipMemcpyDeviceToHost, ctx->stream));
        std::vector<int32_t> ord;
        if (d_ord) {
            ord.resize(n_copy);
            HIPCHK(ctx, hipMemcpyAsync(ord.data(), d_ord, n_copy * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        if (d_ord && !std::is_sorted(ord.begin(), ord.end())) {
            // BAM: rows come out in start order; hand them back in the order pysam iterates them
            std::vector<int32_t> perm(n_copy);
            for (int64_t k = 0; k < n_copy; ++k) perm[k] = (int32_t)k;
            std::sort(perm.begin(), perm.end(), [&](int32_t x, int32_t y) { return ord[x] < ord[y]; });
            auto reorder = [&](auto* col) {
                if (!col) return;
                std::vector<std::remove_pointer_t<decltype(col)>> tmp(col, col + n_copy);
                for (int64_t k = 0; k < n_copy; ++k) col[k] = tmp[perm[k]];
            };
            reorder(len_out);
            reorder(start_out);
            reorder(end_out);
            reorder(mapq_out);
            reorder(strand_out);
        }
    }
    return FTK_OK;
}

int ftk_frag_lengths(ftk_ctx* ctx, int contig_id, int32_t w_start, int32_t w_end, const ftk_filter* f,
                     int32_t* len_out, int64_t cap, int64_t* n_out) {
    return select_common(ctx, contig_id, w_start, w_end, f, len_out, nullptr, nullptr, nullptr, nullptr, cap, n_out);
}

int ftk_frag_select(ftk_ctx* ctx, int contig_id, int32_t w_start, int32_t w_end, const ftk_filter* f,
                    int32_t* start_out, int32_t* end_out, uint8_t* mapq_out, uint8_t* strand_out, int64_t cap,
                    int64_t* n_out) {
    return select_common(ctx, contig_id, w_start, w_end, f, nullptr, start_out, end_out, mapq_out, strand_out, cap,
                         n_out);
}

static int wps_params(ftk_ctx* ctx, const ContigData& c, int64_t chrom_size, int32_t window_size, int32_t min_len,
                      int32_t max_len, int32_t mapq_min, WpsParams* p) {
    if (window_size <= 0 || window_size > (1 << 20)) return fail(ctx, FTK_ERR_INVALID, "window_size out of range");
    if (max_len < 0 || max_len > (1 << 28)) return fail(ctx, FTK_ERR_INVALID, "max_len out of range");
    if (chrom_size < 0) return fail(ctx, FTK_ERR_INVALID, "chrom_size must be >= 0");
    p->chrom_size = chrom_size;
    p->odd = window_size & 1;
    if (p->odd) {
        p->hl = p->hr = (window_size - 1) / 2;
    } else {
        p->hl = window_size / 2;
        p->hr = window_size / 2 - 1;
    }
    p->min_len = min_len < 0 ? 0 : min_len;
    p->max_len = max_len;
    p->mapq_min = mapq_min;
    p->lmax = std::max(0, std::min(max_len, c.max_len));
    // Scores are written once and not re-read by this library: non-temporal stores keep them from
    // displacing the fragment columns in the Infinity Cache and from leaving 256 MB of dirty lines for
    // the next kernel to wait on (measured: -8 % step time, +2 % WPS rate).  FTK_WPS_NT=0 turns it off.
    static const int nt_store = getenv("FTK_WPS_NT") ? atoi(getenv("FTK_WPS_NT")) : 1;
    p->nt_store = nt_store;
    // XCD-contiguous tile ranges would let neighbouring tiles share halo fragments in one L2, but they
    // measured SLOWER (0.79 vs 0.835 of peak): eight far-apart write streams load the HBM channels less
    // evenly than the interleaved order.  Kept as an experiment switch (FTK_WPS_XCD=1), default off.
    static const int xcd_remap = getenv("FTK_WPS_XCD") ? atoi(getenv("FTK_WPS_XCD")) : 0;
    p->xcd_remap = xcd_remap;
    return FTK_OK;
}

#include "ftk_api_perbase.inc"
#include "ftk_api_ref.inc"
#include "ftk_api_comm.inc"
