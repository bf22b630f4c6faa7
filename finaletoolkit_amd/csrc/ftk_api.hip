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
    int rc = stage_in(ctx, w_start, n_win, b_ws, &wc->d_ws);
    if (rc) return rc;
    rc = stage_in(ctx, w_end, n_win, b_we, &wc->d_we);
    if (rc) return rc;
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

// ---- per-base scores to the host on a narrow wire -------------------------------------------------------------------
// The reference's WPS is int64 per base (frag/_wps.py:176-188): 2 GB for chr1, 24.8 GB for a genome - and a device -> host
// copy of it runs at the link's 55 GB/s whatever the kernels do (0.45 s of the 0.54 s a genome's every-feature run takes).
// The VALUES are small (a score is bounded by the fragments over a base: a few hundred at 60x), so they cross the link
// as int16 - a quarter of the bytes - and the host threads widen them into the caller's int64 array while the next
// chunk is on its way.  A score that does not fit (checked on the device, before anything is copied) sends the whole
// result the plain way.  FTK_WPS_NARROW_WIRE=0 keeps the plain copy.
namespace {
constexpr int64_t kNarrowChunk = int64_t(1) << 24;     // scores per chunk (32 MB on the wire, 128 MB widened)
constexpr int64_t kNarrowMin = int64_t(1) << 22;       // shorter results take the plain copy

__global__ __launch_bounds__(256) void narrow_i16_kernel(const int64_t* __restrict__ in, int16_t* __restrict__ out, int64_t n,
                                                         int* __restrict__ misfit) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x * 8;
    bool bad = false;
    for (int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 8 <= n) {
            typedef long long ll2 __attribute__((ext_vector_type(2)));
            const ll2* p = reinterpret_cast<const ll2*>(in + i);  // (in and out are 16-byte aligned, i % 8 == 0)
            long long v[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const ll2 t = __builtin_nontemporal_load(p + k);
                v[2 * k] = t.x;
                v[2 * k + 1] = t.y;
            }
            unsigned w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                bad |= v[2 * k] != (long long)(short)v[2 * k] || v[2 * k + 1] != (long long)(short)v[2 * k + 1];
                w[k] = ((unsigned)v[2 * k] & 0xffffu) | ((unsigned)v[2 * k + 1] << 16);
            }
            *reinterpret_cast<uint4*>(out + i) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
            for (int64_t j = i; j < n; ++j) {
                const long long v = in[j];
                bad |= v != (long long)(short)v;
                out[j] = (int16_t)v;
            }
        }
    }
    if (__ballot(bad) && (threadIdx.x & 63) == 0) atomicOr(misfit, 1);
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) void widen_avx2(const int16_t* src, int64_t* dst, size_t a, size_t b) {
    size_t i = a;
    for (; i < b && ((uintptr_t)(dst + i) & 31u); ++i) dst[i] = src[i];
    for (; i + 16 <= b; i += 16) {  // 16 scores: one 32-byte load, four sign extensions, four streaming stores
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i));
        const __m128i lo = _mm256_castsi256_si128(v), hi = _mm256_extracti128_si256(v, 1);
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), _mm256_cvtepi16_epi64(lo));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 4), _mm256_cvtepi16_epi64(_mm_srli_si128(lo, 8)));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 8), _mm256_cvtepi16_epi64(hi));
        _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 12), _mm256_cvtepi16_epi64(_mm_srli_si128(hi, 8)));
    }
    for (; i < b; ++i) dst[i] = src[i];
    _mm_sfence();
}

#endif

void widen_i16(const int16_t* src, int64_t* dst, size_t n, int nt) {
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
#endif
    nt = std::max(1, std::min(nt, (int)(n >> 18) + 1));
    ftk_host::parallel_run_results(nt, [&](int t) {
        const size_t a = n * (size_t)t / (size_t)nt, b = n * (size_t)(t + 1) / (size_t)nt;
#if defined(__x86_64__)
        if (avx2) { widen_avx2(src, dst, a, b); return; }
#endif
        for (size_t i = a; i < b; ++i) dst[i] = src[i];
    });
}

// d_scores[0, n) (device, int64, complete on ctx->stream) -> host_out[0, n); FTK_OK, or a HIP error.  *done = false:
// nothing was copied (scores that do not fit 16 bits, no page-locked staging): the caller copies the plain way.
int copy_scores_narrow(ftk_ctx* ctx, const int64_t* d_scores, int16_t* d_narrow, int* d_misfit, int64_t n, int64_t* host_out,
                       bool* done) {
    *done = false;
    for (int k = 0; k < 2; ++k) {
        // (page-locked blocks of the library's result cache: a ctx that is destroyed hands them back and the next one
        // finds them there - pinning 64 MB anew for every engine cost the small file legs 4-5 ms)
        if (!ctx->narrow_stage[k] && ftk_host_alloc(kNarrowChunk * 2, &ctx->narrow_stage[k]) != FTK_OK) {
            ctx->narrow_stage[k] = nullptr;
            return FTK_OK;
        }
        if (!ctx->narrow_done[k]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->narrow_done[k], hipEventDisableTiming));
    }
    HIPCHK(ctx, hipMemsetAsync(d_misfit, 0, sizeof(int), ctx->stream));
    const int blocks = (int)std::min<int64_t>((n / 8 + 255) / 256 + 1, 8192);
    hipLaunchKernelGGL(narrow_i16_kernel, dim3(blocks), dim3(256), 0, ctx->stream, d_scores, d_narrow, n, d_misfit);
    HIPCHK(ctx, hipGetLastError());
    int misfit = 0;
    HIPCHK(ctx, hipMemcpyAsync(&misfit, d_misfit, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (misfit) return FTK_OK;
    // FTK_WPS_TIMING=<ms>: a call that takes longer says where (stderr) - kernels, waits for chunks, widening
    static const double slow_ms = getenv("FTK_WPS_TIMING") ? atof(getenv("FTK_WPS_TIMING")) : 0.0;
    const auto t_start = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point t) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
    };
    double t_wait = 0, t_widen = 0, widen_max = 0;
    const int nt = ftk_host::default_threads();
    const int64_t n_chunks = (n + kNarrowChunk - 1) / kNarrowChunk;
    for (int64_t c = 0; c <= n_chunks; ++c) {
        if (c < n_chunks) {  // chunk c on its way (its staging buffer's previous chunk, c - 2, was widened in the last turn)
            const int64_t a = c * kNarrowChunk, m = std::min(kNarrowChunk, n - a);
            HIPCHK(ctx, hipMemcpyAsync(ctx->narrow_stage[c & 1], d_narrow + a, (size_t)m * 2, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipEventRecord(ctx->narrow_done[c & 1], ctx->stream));
        }
        if (c > 0) {  // ... while the host threads widen chunk c - 1
            const int64_t a = (c - 1) * kNarrowChunk, m = std::min(kNarrowChunk, n - a);
            const auto t0 = std::chrono::steady_clock::now();
            HIPCHK(ctx, hipEventSynchronize(ctx->narrow_done[(c - 1) & 1]));
            const auto t1 = std::chrono::steady_clock::now();
            widen_i16(static_cast<const int16_t*>(ctx->narrow_stage[(c - 1) & 1]), host_out + a, (size_t)m, nt);
            const double w = ms_since(t1);
            t_wait += std::chrono::duration<double, std::milli>(t1 - t0).count();
            t_widen += w;
            widen_max = std::max(widen_max, w);
        }
    }
    if (slow_ms > 0 && ms_since(t_start) > slow_ms)
        fprintf(stderr, "[ftk_wps] narrow copy of %lld scores: %.1f ms (waiting for chunks %.1f, widening %.1f - slowest of %lld chunks %.1f - on %d threads)\n",
                (long long)n, ms_since(t_start), t_wait, t_widen, (long long)n_chunks, widen_max, nt);
    *done = true;
    return FTK_OK;
}
}  // namespace

int ftk_wps(ftk_ctx* ctx, int contig_id, int64_t start, int64_t stop, int64_t chrom_size, int32_t window_size,
            int32_t min_len, int32_t max_len, int32_t mapq_min, int64_t* wps_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    ContigData* c;
    int rc = get_contig(ctx, contig_id, &c);
    if (rc) return rc;
    WpsParams p{};
    if ((rc = wps_params(ctx, *c, chrom_size, window_size, min_len, max_len, mapq_min, &p))) return rc;
    if (stop <= start) return FTK_OK;  // degenerate interval: empty result (frag/_wps.py:145-152)
    if (start < -(1LL << 30) || stop > (1LL << 31)) return fail(ctx, FTK_ERR_INVALID, "interval out of range");
    if (!wps_out) return fail(ctx, FTK_ERR_INVALID, "wps_out is NULL");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int64_t n_pos = stop - start;
    static const double slow_ms = getenv("FTK_WPS_TIMING") ? atof(getenv("FTK_WPS_TIMING")) : 0.0;
    const auto t_call = std::chrono::steady_clock::now();
    struct Slow {
        double limit;
        std::chrono::steady_clock::time_point t0;
        long long n;
        double reserve_ms = 0, kernel_ms = 0;
        ~Slow() {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (limit > 0 && ms > limit)
                fprintf(stderr, "[ftk_wps] %lld positions: %.1f ms (scratch %.1f, launch %.1f)\n", n, ms, reserve_ms, kernel_ms);
        }
    } slow{slow_ms, t_call, (long long)n_pos};
    const bool out_dev = is_device_ptr(wps_out);
    static const bool narrow_env = !(getenv("FTK_WPS_NARROW_WIRE") && atoi(getenv("FTK_WPS_NARROW_WIRE")) == 0);
    const bool narrow = !out_dev && narrow_env && n_pos >= kNarrowMin;
    const size_t wide_bytes = align_up((size_t)n_pos * 8);
    if (!out_dev && (rc = reserve_scratch(ctx, wide_bytes + (narrow ? align_up((size_t)n_pos * 2) + 256 : 0)))) return rc;
    slow.reserve_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count();
    int64_t* d_out = out_dev ? wps_out : (int64_t*)ctx->scratch;
    p.start = start;
    p.stop = stop;
    const int64_t n_tiles = (n_pos + kWpsTile - 1) / kWpsTile;
    launch_wps(ctx->stream, c->v, p, n_tiles, nullptr, nullptr, nullptr, nullptr, nullptr, d_out);
    HIPCHK(ctx, hipGetLastError());
    slow.kernel_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count() - slow.reserve_ms;
    if (!out_dev) {
        bool done = false;
        if (narrow) {
            char* tail = (char*)ctx->scratch + wide_bytes;
            if ((rc = copy_scores_narrow(ctx, d_out, (int16_t*)tail, (int*)(tail + align_up((size_t)n_pos * 2)), n_pos, wps_out, &done)))
                return rc;
        }
        if (!done) {
            HIPCHK(ctx, hipMemcpyAsync(wps_out, d_out, n_pos * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    return FTK_OK;
}

// ftk_window_features followed by ftk_wps on the same contig, as ONE launch when the request takes the FAST block
// path (grid = the feature blocks, then the WPS tiles: feat_then_wps_kernel); otherwise the two launches.
int ftk_window_features_wps(ftk_ctx* ctx, int contig_id, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                            const ftk_filter* f, int64_t* count_out, int32_t len_lo, int32_t n_bins, uint32_t* hist_out,
                            int64_t* overflow_out, int32_t delfi_mapq_min, const int32_t* bl_start, const int32_t* bl_end,
                            int64_t n_bl, const ftk_gaps* gaps, int64_t* short_out, int64_t* long_out, int64_t start,
                            int64_t stop, int64_t chrom_size, int32_t window_size, int32_t min_len, int32_t max_len,
                            int32_t mapq_min, int64_t* wps_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    ContigData* c;
    int rc = get_contig(ctx, contig_id, &c);
    if (rc) return rc;
    WpsTail tail{};
    if ((rc = wps_params(ctx, *c, chrom_size, window_size, min_len, max_len, mapq_min, &tail.p))) return rc;
    const bool wps_ok = stop > start && start >= -(1LL << 30) && stop <= (1LL << 31) && wps_out && n_win > 0;
    // scores bound for host memory: the merged launch writes them to the head of the ctx scratch and they cross the link
    // like ftk_wps' (16 bits per score when they fit)
    const bool host_wps = wps_ok && !is_device_ptr(wps_out);
    const int64_t n_pos = stop - start;
    static const bool narrow_env = !(getenv("FTK_WPS_NARROW_WIRE") && atoi(getenv("FTK_WPS_NARROW_WIRE")) == 0);
    const bool narrow = host_wps && narrow_env && n_pos >= kNarrowMin;
    const size_t wide_bytes = host_wps ? align_up((size_t)n_pos * 8) : 0;
    const size_t prefix = host_wps ? wide_bytes + (narrow ? align_up((size_t)n_pos * 2) + 256 : 0) : 0;
    FeatCall fc;
    fc.f = f;
    fc.count_out = count_out;
    fc.hist_out = hist_out;
    fc.overflow_out = overflow_out;
    fc.len_lo = len_lo;
    fc.n_bins = n_bins;
    fc.delfi = short_out || long_out;
    fc.mapq_min = delfi_mapq_min;
    fc.bl_start = bl_start;
    fc.bl_end = bl_end;
    fc.n_bl = n_bl;
    fc.gaps = gaps;
    fc.short_out = short_out;
    fc.long_out = long_out;
    bool merged = false;
    if (wps_ok) {
        tail.p.start = start;
        tail.p.stop = stop;
        tail.n_tiles = (stop - start + kWpsTile - 1) / kWpsTile;
        tail.out = wps_out;  // (host_wps: replaced by the scratch base inside features_common)
    }
    if ((rc = features_common(ctx, contig_id, w_start, w_end, n_win, fc, wps_ok ? &tail : nullptr, &merged, prefix))) return rc;
    if (!merged) return ftk_wps(ctx, contig_id, start, stop, chrom_size, window_size, min_len, max_len, mapq_min, wps_out);
    if (host_wps) {
        int64_t* d_out = (int64_t*)ctx->scratch;
        bool done = false;
        if (narrow) {
            char* nw = (char*)ctx->scratch + wide_bytes;
            if ((rc = copy_scores_narrow(ctx, d_out, (int16_t*)nw, (int*)(nw + align_up((size_t)n_pos * 2)), n_pos, wps_out, &done)))
                return rc;
        }
        if (!done) {
            HIPCHK(ctx, hipMemcpyAsync(wps_out, d_out, (size_t)n_pos * 8, hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
        }
    }
    return FTK_OK;
}

int ftk_wps_async(ftk_ctx* ctx, int contig_id, int64_t start, int64_t stop, int64_t chrom_size, int32_t window_size,
                  int32_t min_len, int32_t max_len, int32_t mapq_min, int64_t* wps_out_host, int* token_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    ContigData* c;
    int rc = get_contig(ctx, contig_id, &c);
    if (rc) return rc;
    WpsParams p{};
    if ((rc = wps_params(ctx, *c, chrom_size, window_size, min_len, max_len, mapq_min, &p))) return rc;
    if (!token_out) return fail(ctx, FTK_ERR_INVALID, "token_out is NULL");
    *token_out = -1;
    if (stop <= start) return FTK_OK;  // degenerate interval: empty result, nothing to wait for
    if (start < -(1LL << 30) || stop > (1LL << 31)) return fail(ctx, FTK_ERR_INVALID, "interval out of range");
    if (!wps_out_host || is_device_ptr(wps_out_host)) return fail(ctx, FTK_ERR_INVALID, "wps_out_host must be a host array");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    if (!ctx->copy_stream) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->a_kernel_done[k], hipEventDisableTiming));
            HIPCHK(ctx, hipEventCreateWithFlags(&ctx->a_copy_done[k], hipEventDisableTiming));
        }
    }
    const int k = ctx->a_issued & 1;  // tokens count up; a token's buffer is its lowest bit
    if (ctx->a_pending[k]) {  // the buffer's previous result is still on its way to the host
        HIPCHK(ctx, hipEventSynchronize(ctx->a_copy_done[k]));
        ctx->a_pending[k] = false;
    }
    const int64_t n_pos = stop - start;
    const size_t need = align_up((size_t)n_pos * 8);
    if (need > ctx->abuf_bytes[k]) {
        if (ctx->abuf[k]) HIPCHK(ctx, hipFree(ctx->abuf[k]));
        ctx->abuf[k] = nullptr;
        ctx->abuf_bytes[k] = 0;
        HIPCHK(ctx, hipMalloc(&ctx->abuf[k], need));
        ctx->abuf_bytes[k] = need;
    }
    p.start = start;
    p.stop = stop;
    const int64_t n_tiles = (n_pos + kWpsTile - 1) / kWpsTile;
    launch_wps(ctx->stream, c->v, p, n_tiles, nullptr, nullptr, nullptr, nullptr, nullptr, (int64_t*)ctx->abuf[k]);
    HIPCHK(ctx, hipGetLastError());
    HIPCHK(ctx, hipEventRecord(ctx->a_kernel_done[k], ctx->stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->copy_stream, ctx->a_kernel_done[k], 0));
    HIPCHK(ctx, hipMemcpyAsync(wps_out_host, ctx->abuf[k], (size_t)n_pos * 8, hipMemcpyDeviceToHost, ctx->copy_stream));
    HIPCHK(ctx, hipEventRecord(ctx->a_copy_done[k], ctx->copy_stream));
    ctx->a_pending[k] = true;
    ctx->a_token[k] = ctx->a_issued;
    *token_out = ctx->a_issued;
    ctx->a_issued = ctx->a_issued == INT32_MAX ? (ctx->a_issued & 1) ^ 1 : ctx->a_issued + 1;  // (wraps with the parity kept)
    return FTK_OK;
}

int ftk_result_wait(ftk_ctx* ctx, int token) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    if (token < 0) return FTK_OK;  // an empty result
    const int k = token & 1;
    if (ctx->a_token[k] < 0 || (ctx->a_token[k] != token && ctx->a_token[1 - k] != token && token >= ctx->a_issued))
        return fail(ctx, FTK_ERR_INVALID, "result token %d was never handed out", token);
    // a token older than the one in its buffer names a result that is complete: the call that reused the buffer waited
    // for its copy first
    if (ctx->a_token[k] == token && ctx->a_pending[k]) {
        HIPCHK(ctx, hipSetDevice(ctx->device));
        HIPCHK(ctx, hipEventSynchronize(ctx->a_copy_done[k]));
        ctx->a_pending[k] = false;
    }
    return FTK_OK;
}

int ftk_wps_intervals(ftk_ctx* ctx, int contig_id, const int64_t* iv_start, const int64_t* iv_stop, int64_t n_iv,
                      const int64_t* out_offset, int64_t chrom_size, int32_t window_size, int32_t min_len,
                      int32_t max_len, int32_t mapq_min, int64_t* wps_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    ContigData* c;
    int rc = get_contig(ctx, contig_id, &c);
    if (rc) return rc;
    WpsParams p{};
    if ((rc = wps_params(ctx, *c, chrom_size, window_size, min_len, max_len, mapq_min, &p))) return rc;
    if (n_iv < 0 || n_iv > INT32_MAX) return fail(ctx, FTK_ERR_INVALID, "n_iv out of range");
    if (n_iv == 0) return FTK_OK;
    if (!iv_start || !iv_stop || !out_offset || !wps_out) return fail(ctx, FTK_ERR_INVALID, "NULL argument");
    if (is_device_ptr(iv_start) || is_device_ptr(iv_stop) || is_device_ptr(out_offset))
        return fail(ctx, FTK_ERR_INVALID, "interval arrays must be host arrays");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    std::vector<int32_t> tile_iv, tile_k;
    int64_t total_out = 0;
    for (int64_t i = 0; i < n_iv; ++i) {
        int64_t len = iv_stop[i] - iv_start[i];
        if (len <= 0) continue;
        if (iv_start[i] < -(1LL << 30) || iv_stop[i] > (1LL << 31)) return fail(ctx, FTK_ERR_INVALID, "interval out of range");
        if (out_offset[i] < 0) return fail(ctx, FTK_ERR_INVALID, "negative output offset");
        total_out = std::max(total_out, out_offset[i] + len);
        int64_t nt = (len + kWpsTile - 1) / kWpsTile;
        for (int64_t k = 0; k < nt; ++k) {
            tile_iv.push_back((int32_t)i);
            tile_k.push_back((int32_t)k);
        }
    }
    const size_t n_tiles = tile_iv.size();
    if (n_tiles == 0) return FTK_OK;
    if (n_tiles > (size_t)INT32_MAX) return fail(ctx, FTK_ERR_INVALID, "too many WPS tiles in one call");
    const bool out_dev = is_device_ptr(wps_out);
    size_t need = 3 * align_up(n_iv * 8) + 2 * align_up(n_tiles * 4) + (out_dev ? 0 : align_up(total_out * 8));
    if ((rc = reserve_scratch(ctx, need))) return rc;
    Arena a(ctx);
    int64_t* d_s = a.take<int64_t>(n_iv);
    int64_t* d_e = a.take<int64_t>(n_iv);
    int64_t* d_o = a.take<int64_t>(n_iv);
    int32_t* d_ti = a.take<int32_t>(n_tiles);
    int32_t* d_tk = a.take<int32_t>(n_tiles);
    int64_t* d_out = out_dev ? wps_out : a.take<int64_t>(total_out);
    HIPCHK(ctx, hipMemcpyAsync(d_s, iv_start, n_iv * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_e, iv_stop, n_iv * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_o, out_offset, n_iv * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_ti, tile_iv.data(), n_tiles * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_tk, tile_k.data(), n_tiles * 4, hipMemcpyHostToDevice, ctx->stream));
    launch_wps(ctx->stream, c->v, p, (int64_t)n_tiles, d_s, d_e, d_o, d_ti, d_tk, d_out);
    HIPCHK(ctx, hipGetLastError());
    if (!out_dev) HIPCHK(ctx, hipMemcpyAsync(wps_out, d_out, total_out * 8, hipMemcpyDeviceToHost, ctx->stream));
    // tile descriptor vectors are pageable staging: wait before they go away
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FTK_OK;
}

int ftk_wps_window_features(ftk_ctx* ctx, int contig_id, int64_t start, int64_t stop, int64_t chrom_size,
                            int32_t window_size, int32_t min_len, int32_t max_len, int32_t mapq_min, int64_t* wps_out,
                            int32_t win_start, int32_t win_len, int32_t n_win, const ftk_filter* f, int64_t* count_out,
                            int32_t len_lo, int32_t n_bins, uint32_t* hist_out, int64_t* overflow_out,
                            int32_t delfi_mapq_min, const int32_t* bl_start, const int32_t* bl_end, int64_t n_bl,
                            const ftk_gaps* gaps, int64_t* short_out, int64_t* long_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    ContigData* c;
    int rc = get_contig(ctx, contig_id, &c);
    if (rc) return rc;
    WpsParams p{};
    if ((rc = wps_params(ctx, *c, chrom_size, window_size, min_len, max_len, mapq_min, &p))) return rc;
    const bool ch = count_out || hist_out, df = short_out || long_out;
    if (!wps_out || (!ch && !df)) return fail(ctx, FTK_ERR_INVALID, "needs wps_out and at least one feature output");
    if (ch && (!f || f->policy != FTK_POLICY_MIDPOINT))
        return fail(ctx, FTK_ERR_INVALID, "the fused pass needs a midpoint-policy filter");
    if (ch && (rc = check_filter(ctx, f, *c))) return rc;
    // fetch semantics as everywhere else: read1 overlap (io/alignment.py:245) on a contig with read1 columns unless
    // the filter asks for the tabix rule
    const bool bam = c->v.r1_start != nullptr && (!ch || f->fetch_mode == FTK_FETCH_BAM_READ1);
    if (df && (!short_out || !long_out)) return fail(ctx, FTK_ERR_INVALID, "NULL DELFI output pointer");
    if (hist_out && (n_bins <= 0 || n_bins > 8192 || !overflow_out))
        return fail(ctx, FTK_ERR_INVALID, "histogram needs n_bins in [1, 8192] and overflow_out");
    if (n_win <= 0 || win_start < 0 || win_len < kWpsTile + c->max_len || (int64_t)win_start + (int64_t)n_win * win_len > (1LL << 31))
        return fail(ctx, FTK_ERR_INVALID, "bins must be at least %d bp long (tile + longest fragment) and end below 2^31",
                    kWpsTile + c->max_len);
    if (start > 0 || stop < (int64_t)c->max_end || stop <= start)
        return fail(ctx, FTK_ERR_INVALID, "the WPS interval must cover every fragment of the contig (start <= 0, stop >= %d)",
                    c->max_end);
    if (start < -(1LL << 30) || stop > (1LL << 31)) return fail(ctx, FTK_ERR_INVALID, "interval out of range");
    if (n_bl < 0 || (n_bl > 0 && (!bl_start || !bl_end)) || is_device_ptr(bl_start))
        return fail(ctx, FTK_ERR_INVALID, "bad blacklist arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    FusedParams F{};
    F.win_start = win_start;
    F.win_len = win_len;
    F.n_win = n_win;
    F.bam = bam;
    if (ch) {
        F.ch_q = std::min(std::max(f->mapq_min, 0), 256);
        F.ch_min = f->min_len < 0 ? 0 : std::min(f->min_len, 1 << 30);
        F.ch_max = f->max_len < 0 ? (1 << 30) : std::min(f->max_len, 1 << 30);
    }
    F.do_cov = count_out != nullptr;
    F.do_hist = hist_out != nullptr;
    F.len_lo = len_lo;
    F.n_bins = n_bins;
    F.do_delfi = df;
    F.df_q = std::min(std::max(delfi_mapq_min, 0), 256);
    ftk_gaps g{};
    if (gaps) g = *gaps;
    if (g.has_gaps && (g.n_telo < 0 || g.n_telo > FTK_MAX_TELOMERES))
        return fail(ctx, FTK_ERR_INVALID, "at most %d telomere intervals per contig are supported", FTK_MAX_TELOMERES);
    int gc[4];
    gap_constants(g, gc);
    F.cen0 = gc[0]; F.cen1 = gc[1]; F.tel0 = gc[2]; F.tel1 = gc[3];
    if (df && n_bl > 0) {  // per-bin blacklist CSR, cached like ftk_delfi_counts' (same key: the bins' arrays)
        std::vector<int32_t> ws(n_win), we(n_win);
        for (int k = 0; k < n_win; ++k) { ws[k] = win_start + k * win_len; we[k] = ws[k] + win_len; }
        DelfiMeta* meta = nullptr;
        if ((rc = get_delfi_meta(ctx, contig_id, ws.data(), we.data(), n_win, bl_start, bl_end, n_bl, &meta))) return rc;
        if (meta->n_r) { F.bl_off = meta->d_off; F.bl_r0 = meta->d_r0; F.bl_pm = meta->d_pm; }
    }
    const int64_t n_pos = stop - start;
    const bool w_dev = is_device_ptr(wps_out), c_dev = is_device_ptr(count_out), h_dev = is_device_ptr(hist_out),
               o_dev = is_device_ptr(overflow_out), s_dev = is_device_ptr(short_out), l_dev = is_device_ptr(long_out);
    const size_t hist_elems = hist_out ? (size_t)n_win * (size_t)n_bins : 0;
    if ((rc = reserve_scratch(ctx, (w_dev ? 0 : align_up(n_pos * 8)) + 4 * align_up((size_t)n_win * 8) +
                                       (h_dev ? 0 : align_up(hist_elems * 4)))))
        return rc;
    Arena a(ctx);
    int64_t* d_wps = w_dev ? wps_out : a.take<int64_t>(n_pos);
    F.cov_out = count_out ? (c_dev ? count_out : a.take<int64_t>(n_win)) : nullptr;
    F.hist_out = hist_out ? (h_dev ? hist_out : a.take<uint32_t>(hist_elems)) : nullptr;
    F.over_out = hist_out ? (o_dev ? overflow_out : a.take<int64_t>(n_win)) : nullptr;
    F.short_out = df ? (s_dev ? short_out : a.take<int64_t>(n_win)) : nullptr;
    F.long_out = df ? (l_dev ? long_out : a.take<int64_t>(n_win)) : nullptr;
    // every feature output is accumulated with atomics: start from zero
    if (F.cov_out) HIPCHK(ctx, hipMemsetAsync(F.cov_out, 0, (size_t)n_win * 8, ctx->stream));
    if (F.hist_out) {
        HIPCHK(ctx, hipMemsetAsync(F.hist_out, 0, hist_elems * 4, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(F.over_out, 0, (size_t)n_win * 8, ctx->stream));
    }
    if (df) {
        HIPCHK(ctx, hipMemsetAsync(F.short_out, 0, (size_t)n_win * 8, ctx->stream));
        HIPCHK(ctx, hipMemsetAsync(F.long_out, 0, (size_t)n_win * 8, ctx->stream));
    }
    p.start = start;
    p.stop = stop;
    launch_wps_fused(ctx->stream, c->v, p, (n_pos + kWpsTile - 1) / kWpsTile, F, d_wps);
    HIPCHK(ctx, hipGetLastError());
    bool host_out = false;
    auto back = [&](void* dst, const void* src, size_t bytes, bool dev) -> hipError_t {
        if (!dst || dev) return hipSuccess;
        host_out = true;
        return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream);
    };
    HIPCHK(ctx, back(wps_out, d_wps, (size_t)n_pos * 8, w_dev));
    HIPCHK(ctx, back(count_out, F.cov_out, (size_t)n_win * 8, c_dev));
    HIPCHK(ctx, back(hist_out, F.hist_out, hist_elems * 4, h_dev));
    HIPCHK(ctx, back(hist_out ? overflow_out : nullptr, F.over_out, (size_t)n_win * 8, o_dev));
    HIPCHK(ctx, back(short_out, F.short_out, (size_t)n_win * 8, s_dev));
    HIPCHK(ctx, back(long_out, F.long_out, (size_t)n_win * 8, l_dev));
    if (host_out) HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    return FTK_OK;
}

int ftk_wps_batch(ftk_ctx* ctx, const int32_t* contig_ids, const int64_t* iv_start, const int64_t* iv_stop,
                  const int64_t* chrom_size, const int64_t* out_offset, int64_t n_iv, int32_t window_size,
                  int32_t min_len, int32_t max_len, int32_t mapq_min, int64_t* wps_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    if (n_iv < 0 || n_iv > kBatchMaxItems) return fail(ctx, FTK_ERR_INVALID, "n_iv must be in [0, %d]", kBatchMaxItems);
    if (n_iv == 0) return FTK_OK;
    if (!contig_ids || !iv_start || !iv_stop || !chrom_size || !out_offset || !wps_out)
        return fail(ctx, FTK_ERR_INVALID, "NULL argument");
    const bool out_dev = is_device_ptr(wps_out);
    long long total_out = 0;
    std::vector<WpsItem> host;
    WpsParams p{};
    long long tiles = 0;
    for (int64_t i = 0; i < n_iv; ++i) {
        ContigData* c;
        int rc = get_contig(ctx, contig_ids[i], &c);
        if (rc) return rc;
        WpsParams pi{};
        if ((rc = wps_params(ctx, *c, chrom_size[i], window_size, min_len, max_len, mapq_min, &pi))) return rc;
        const long long len = iv_stop[i] - iv_start[i];
        if (len <= 0) continue;
        if (iv_start[i] < -(1LL << 30) || iv_stop[i] > (1LL << 31) || out_offset[i] < 0)
            return fail(ctx, FTK_ERR_INVALID, "interval %lld out of range", (long long)i);
        if (host.empty()) p = pi;
        if ((c->v.r1_start != nullptr) != (host.empty() ? c->v.r1_start != nullptr : host[0].cv.r1_start != nullptr))
            return fail(ctx, FTK_ERR_INVALID, "a batch cannot mix contigs with and without read1 columns");
        WpsItem it{};
        it.cv = c->v;
        it.start = iv_start[i];
        it.stop = iv_stop[i];
        it.chrom_size = chrom_size[i];
        it.out_off = out_offset[i];
        it.tile_base = tiles;
        it.lmax = pi.lmax;
        host.push_back(it);
        tiles += (len + kWpsTile - 1) / kWpsTile;
        total_out = std::max(total_out, (long long)out_offset[i] + len);
    }
    if (host.empty()) return FTK_OK;
    if (tiles > (long long)INT32_MAX) return fail(ctx, FTK_ERR_INVALID, "too many tiles in one call");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    void* d_items = nullptr;
    int rc = upload_batch_descriptors(ctx, 1, host.data(), host.size() * sizeof(WpsItem), &d_items);
    if (rc) return rc;
    int64_t* d_out = wps_out;
    if (!out_dev) {
        if ((rc = reserve_scratch(ctx, align_up((size_t)total_out * 8)))) return rc;
        d_out = (int64_t*)ctx->scratch;
    }
    launch_wps_batch(ctx->stream, p, (const WpsItem*)d_items, (int)host.size(), tiles, d_out);
    HIPCHK(ctx, hipGetLastError());
    if (!out_dev) {
        HIPCHK(ctx, hipMemcpyAsync(wps_out, d_out, (size_t)total_out * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return FTK_OK;
}

int ftk_cleavage_intervals(ftk_ctx* ctx, int contig_id, const int64_t* iv_start, const int64_t* iv_stop, int64_t n_iv,
                           const int64_t* out_offset, int32_t min_len, int32_t max_len, int32_t mapq_min,
                           double* prop_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    ContigData* c;
    int rc = get_contig(ctx, contig_id, &c);
    if (rc) return rc;
    if (n_iv < 0 || n_iv > INT32_MAX) return fail(ctx, FTK_ERR_INVALID, "n_iv out of range");
    if (n_iv == 0) return FTK_OK;
    if (!iv_start || !iv_stop || !out_offset || !prop_out) return fail(ctx, FTK_ERR_INVALID, "NULL argument");
    if (is_device_ptr(iv_start) || is_device_ptr(iv_stop) || is_device_ptr(out_offset))
        return fail(ctx, FTK_ERR_INVALID, "interval arrays must be host arrays");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    CleaveParams p{};
    p.min_len = min_len < 0 ? INT32_MIN : min_len;
    p.max_len = max_len < 0 ? INT32_MAX : max_len;
    p.mapq_min = mapq_min;
    p.lmax = std::max(0, max_len < 0 ? c->max_len : std::min(max_len, c->max_len));
    std::vector<int32_t> tile_iv, tile_k;
    int64_t total_out = 0;
    for (int64_t i = 0; i < n_iv; ++i) {
        const int64_t len = iv_stop[i] - iv_start[i];
        if (len <= 0) continue;
        if (iv_start[i] < 0 || iv_stop[i] > (1LL << 31)) return fail(ctx, FTK_ERR_INVALID, "interval out of range");
        if (out_offset[i] < 0) return fail(ctx, FTK_ERR_INVALID, "negative output offset");
        total_out = std::max(total_out, out_offset[i] + len);
        for (int64_t k = 0; k < (len + kWpsTile - 1) / kWpsTile; ++k) {
            tile_iv.push_back((int32_t)i);
            tile_k.push_back((int32_t)k);
        }
    }
    const size_t n_tiles = tile_iv.size();
    if (n_tiles == 0) return FTK_OK;
    if (n_tiles > (size_t)INT32_MAX) return fail(ctx, FTK_ERR_INVALID, "too many tiles in one call");
    const bool out_dev = is_device_ptr(prop_out);
    size_t need = 3 * align_up(n_iv * 8) + 2 * align_up(n_tiles * 4) + (out_dev ? 0 : align_up(total_out * 8));
    if ((rc = reserve_scratch(ctx, need))) return rc;
    Arena a(ctx);
    int64_t* d_s = a.take<int64_t>(n_iv);
    int64_t* d_e = a.take<int64_t>(n_iv);
    int64_t* d_o = a.take<int64_t>(n_iv);
    int32_t* d_ti = a.take<int32_t>(n_tiles);
    int32_t* d_tk = a.take<int32_t>(n_tiles);
    double* d_out = out_dev ? prop_out : a.take<double>(total_out);
    HIPCHK(ctx, hipMemcpyAsync(d_s, iv_start, n_iv * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_e, iv_stop, n_iv * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_o, out_offset, n_iv * 8, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_ti, tile_iv.data(), n_tiles * 4, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_tk, tile_k.data(), n_tiles * 4, hipMemcpyHostToDevice, ctx->stream));
    launch_cleavage(ctx->stream, c->v, p, (int64_t)n_tiles, d_s, d_e, d_o, d_ti, d_tk, d_out);
    HIPCHK(ctx, hipGetLastError());
    if (!out_dev) HIPCHK(ctx, hipMemcpyAsync(prop_out, d_out, total_out * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // the tile descriptor vectors are pageable staging
    return FTK_OK;
}

int ftk_cleavage(ftk_ctx* ctx, int contig_id, int64_t start, int64_t stop, int32_t min_len, int32_t max_len,
                 int32_t mapq_min, double* prop_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    if (stop <= start) return FTK_OK;
    ContigData* c;
    int rc = get_contig(ctx, contig_id, &c);
    if (rc) return rc;
    if (start < 0 || stop > (1LL << 31)) return fail(ctx, FTK_ERR_INVALID, "interval out of range");
    if (!prop_out) return fail(ctx, FTK_ERR_INVALID, "NULL argument");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    // ONE interval: its tiles are numbered by the grid (no descriptor arrays to build and upload - for a whole contig
    // 60 000 tiles, half a megabyte of pageable staging and a stream synchronisation per call)
    CleaveParams p{};
    p.start = start;
    p.stop = stop;
    p.min_len = min_len < 0 ? INT32_MIN : min_len;
    p.max_len = max_len < 0 ? INT32_MAX : max_len;
    p.mapq_min = mapq_min;
    p.lmax = std::max(0, max_len < 0 ? c->max_len : std::min(max_len, c->max_len));
    const int64_t n_pos = stop - start, n_tiles = (n_pos + kWpsTile - 1) / kWpsTile;
    const bool out_dev = is_device_ptr(prop_out);
    if (!out_dev && (rc = reserve_scratch(ctx, align_up((size_t)n_pos * 8)))) return rc;
    double* d_out = out_dev ? prop_out : (double*)ctx->scratch;
    launch_cleavage(ctx->stream, c->v, p, n_tiles, nullptr, nullptr, nullptr, nullptr, nullptr, d_out);
    HIPCHK(ctx, hipGetLastError());
    if (!out_dev) {
        HIPCHK(ctx, hipMemcpyAsync(prop_out, d_out, (size_t)n_pos * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return FTK_OK;
}

int ftk_wps_adjust(ftk_ctx* ctx, const double* scores, const int64_t* offsets, int64_t n_iv, int32_t median_window,
                   int use_mean, const double* edge_sub, int32_t savgol_window, const double* savgol_coef,
                   const double* savgol_edge, double* out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    if (n_iv < 0 || n_iv > INT32_MAX) return fail(ctx, FTK_ERR_INVALID, "n_iv out of range");
    if (n_iv == 0) return FTK_OK;
    if (!scores || !offsets || !out) return fail(ctx, FTK_ERR_INVALID, "NULL argument");
    if (is_device_ptr(offsets) || is_device_ptr(edge_sub) || is_device_ptr(savgol_coef) ||
        is_device_ptr(savgol_edge))
        return fail(ctx, FTK_ERR_INVALID, "offsets / edge_sub / savgol arrays must be host arrays");
    const int W = median_window;
    if (W < 2 || (W & 1) || W > kAdjustMaxWindow)
        return fail(ctx, FTK_ERR_INVALID, "median_window must be even and in [2, %d]", kAdjustMaxWindow);
    const int sw = savgol_window;
    if (sw < 0 || (sw > 0 && (!(sw & 1) || !savgol_coef || !savgol_edge)))
        return fail(ctx, FTK_ERR_INVALID, "savgol_window must be odd and come with coef/edge arrays");
    int tile;
    adjust_sort_size(W, &tile);
    // FTK_ADJUST_HIST=0: every interval through the sort kernel (the tests hold the two medians together)
    static const bool use_hist = !(getenv("FTK_ADJUST_HIST") && atoi(getenv("FTK_ADJUST_HIST")) == 0);
    const bool fast = use_hist && !use_mean;
    std::vector<AdjustTile> tiles, fast_tiles;
    for (int64_t i = 0; i < n_iv; ++i) {
        const int64_t len = offsets[i + 1] - offsets[i];
        if (offsets[i] < 0 || len < W)
            return fail(ctx, FTK_ERR_INVALID, "run %lld is shorter than median_window (%lld < %d)", (long long)i,
                        (long long)len, W);
        const int64_t m = len - W;
        if (m > INT32_MAX) return fail(ctx, FTK_ERR_INVALID, "run %lld too long", (long long)i);
        if (sw > 0 && m < sw)
            return fail(ctx, FTK_ERR_INVALID, "run %lld: savgol_window (%d) exceeds the filtered length (%lld)",
                        (long long)i, sw, (long long)m);
        const int64_t out_i = offsets[i] - i * (int64_t)W;
        for (int64_t o0 = 0; o0 < m; o0 += tile) {
            AdjustTile t;
            t.in_base = offsets[i] + o0;
            t.out_base = out_i + o0;
            t.n_out = (int32_t)std::min<int64_t>(tile, m - o0);
            t.o0 = (int32_t)o0;
            t.m = (int32_t)m;
            t.interval = (int32_t)i;
            tiles.push_back(t);
        }
        for (int64_t o0 = 0; fast && o0 < m; o0 += kAdjustFastTile) {
            AdjustTile t;
            t.in_base = offsets[i] + o0;
            t.out_base = out_i + o0;
            t.n_out = (int32_t)std::min<int64_t>(kAdjustFastTile, m - o0);
            t.o0 = (int32_t)o0;
            t.m = (int32_t)m;
            t.interval = (int32_t)i;
            fast_tiles.push_back(t);
        }
    }
    const int64_t total_in = offsets[n_iv], total_out = total_in - n_iv * (int64_t)W;
    if (tiles.empty()) return FTK_OK;
    if (tiles.size() > (size_t)INT32_MAX) return fail(ctx, FTK_ERR_INVALID, "too many tiles in one call");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool in_dev = is_device_ptr(scores), out_dev = is_device_ptr(out);
    const int half = sw / 2;
    size_t need = align_up(tiles.size() * sizeof(AdjustTile)) + align_up(fast_tiles.size() * sizeof(AdjustTile)) + align_up(n_iv * 4) +
                  align_up(n_iv * 8) + align_up((size_t)sw * 8) +
                  align_up((size_t)2 * half * sw * 8) + (in_dev ? 0 : align_up(total_in * 8)) +
                  (out_dev ? 0 : align_up(total_out * 8)) + (sw ? align_up(total_out * 8) : 0);
    int rc = reserve_scratch(ctx, need);
    if (rc) return rc;
    Arena a(ctx);
    AdjustTile* d_tiles = a.take<AdjustTile>(tiles.size());
    AdjustTile* d_fast = fast_tiles.empty() ? nullptr : a.take<AdjustTile>(fast_tiles.size());
    int* d_todo = fast_tiles.empty() ? nullptr : a.take<int>(n_iv);
    double* d_sub = edge_sub ? a.take<double>(n_iv) : nullptr;
    double* d_coef = sw ? a.take<double>(sw) : nullptr;
    double* d_edge = sw ? a.take<double>((size_t)2 * half * sw) : nullptr;
    const double* d_in = scores;
    if (!in_dev) {
        double* b = a.take<double>(total_in);
        HIPCHK(ctx, hipMemcpyAsync(b, scores, total_in * 8, hipMemcpyHostToDevice, ctx->stream));
        d_in = b;
    }
    double* d_out = out_dev ? out : a.take<double>(total_out);
    double* d_adj = sw ? a.take<double>(total_out) : d_out;
    HIPCHK(ctx, hipMemcpyAsync(d_tiles, tiles.data(), tiles.size() * sizeof(AdjustTile), hipMemcpyHostToDevice,
                               ctx->stream));
    if (d_fast)
        HIPCHK(ctx, hipMemcpyAsync(d_fast, fast_tiles.data(), fast_tiles.size() * sizeof(AdjustTile), hipMemcpyHostToDevice, ctx->stream));
    if (d_sub) HIPCHK(ctx, hipMemcpyAsync(d_sub, edge_sub, n_iv * 8, hipMemcpyHostToDevice, ctx->stream));
    if (sw) {
        HIPCHK(ctx, hipMemcpyAsync(d_coef, savgol_coef, (size_t)sw * 8, hipMemcpyHostToDevice, ctx->stream));
        if (half)
            HIPCHK(ctx, hipMemcpyAsync(d_edge, savgol_edge, (size_t)2 * half * sw * 8, hipMemcpyHostToDevice,
                                       ctx->stream));
    }
    launch_adjust_filter(ctx->stream, d_in, d_tiles, (int)tiles.size(), d_sub, W, use_mean, d_adj, d_fast, (int)fast_tiles.size(), d_todo,
                         (int)n_iv);
    if (sw) launch_savgol(ctx->stream, d_adj, d_tiles, (int)tiles.size(), d_coef, d_edge, sw, d_out);
    HIPCHK(ctx, hipGetLastError());
    if (!out_dev) HIPCHK(ctx, hipMemcpyAsync(out, d_out, total_out * 8, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // tiles / coefficient arrays are pageable staging
    return FTK_OK;
}

namespace {

constexpr size_t kRefStageBytes = size_t(16) << 20;
constexpr size_t kRefPoolMax = 4;

// a device block of at least `bytes` for a reference image: the smallest pooled one that fits, else a fresh one
int ref_block_take(ftk_ctx* ctx, size_t bytes, void** out, size_t* cap) {
    int best = -1;
    for (size_t i = 0; i < ctx->ref_pool.size(); ++i)
        if (ctx->ref_pool[i].second >= bytes && (best < 0 || ctx->ref_pool[i].second < ctx->ref_pool[best].second)) best = (int)i;
    if (best >= 0) {
        *out = ctx->ref_pool[best].first;
        *cap = ctx->ref_pool[best].second;
        ctx->ref_pool.erase(ctx->ref_pool.begin() + best);
        return FTK_OK;
    }
    const size_t want = align_up(bytes + bytes / 8, 1 << 20);  // a little slack: the next contig is often slightly larger
    HIPCHK(ctx, hipMalloc(out, want));
    *cap = want;
    return FTK_OK;
}

void ref_block_give(ftk_ctx* ctx, void* d, size_t cap) {
    if (!d) return;
    ctx->ref_pool.emplace_back(d, cap);
    while (ctx->ref_pool.size() > kRefPoolMax) {  // keep the largest blocks
        size_t k = 0;
        for (size_t i = 1; i < ctx->ref_pool.size(); ++i)
            if (ctx->ref_pool[i].second < ctx->ref_pool[k].second) k = i;
        (void)hipFree(ctx->ref_pool[k].first);
        ctx->ref_pool.erase(ctx->ref_pool.begin() + (long)k);
    }
}

int ref_drop(ftk_ctx* ctx, int ref_id) {  // an image about to be replaced / released: its blocks go back to the pool
    auto it = ctx->refs.find(ref_id);
    if (it == ctx->refs.end()) return FTK_OK;
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));  // launches that read it have finished
    ref_block_give(ctx, it->second.d, (size_t)it->second.cap);
    if (it->second.d_nblk) (void)hipFree(it->second.d_nblk);
    ctx->refs.erase(it);
    return FTK_OK;
}

}  // namespace

int ftk_ref_upload(ftk_ctx* ctx, int ref_id, const uint8_t* image, int64_t n_bytes, int kind) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    if (n_bytes < 0 || (n_bytes > 0 && !image) || (kind != FTK_REF_FASTA_TEXT && kind != FTK_REF_2BIT))
        return fail(ctx, FTK_ERR_INVALID, "bad reference image arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = ref_drop(ctx, ref_id);
    if (rc) return rc;
    ftk_ctx::RefImage r;
    r.bytes = n_bytes;
    r.kind = kind;
    size_t cap = 0;
    if ((rc = ref_block_take(ctx, (size_t)n_bytes + 32, &r.d, &cap))) return rc;  // padded: 16-byte loads never leave the block
    r.cap = (int64_t)cap;
    // on the ctx stream, one wait for THAT stream at the end (a synchronous hipMemcpy waits for every transfer the
    // device has in flight)
    hipError_t e = hipMemsetAsync((char*)r.d + n_bytes, 0, 32, ctx->stream);
    if (e == hipSuccess && n_bytes)
        e = hipMemcpyAsync(r.d, image, n_bytes, is_device_ptr(image) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(r.d);
        return fail(ctx, FTK_ERR_HIP, "reference upload failed: %s", hipGetErrorString(e));
    }
    ctx->refs[ref_id] = r;
    return FTK_OK;
}

int ftk_ref_upload_file(ftk_ctx* ctx, int ref_id, const char* path, int64_t file_offset, int64_t n_bytes, int kind) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    if (!path || file_offset < 0 || n_bytes < 0 || (kind != FTK_REF_FASTA_TEXT && kind != FTK_REF_2BIT))
        return fail(ctx, FTK_ERR_INVALID, "bad reference image arguments");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const int fd = open(path, O_RDONLY);
    if (fd < 0) return fail(ctx, FTK_ERR_IO, "cannot open %s: %s", path, strerror(errno));
    struct Closer { int fd; ~Closer() { close(fd); } } closer{fd};
    int rc = ref_drop(ctx, ref_id);
    if (rc) return rc;
    for (int k = 0; k < 2; ++k) {
        if (!ctx->ref_stage[k]) HIPCHK(ctx, hipHostMalloc(&ctx->ref_stage[k], kRefStageBytes, hipHostMallocDefault));
        if (!ctx->ref_stage_done[k]) HIPCHK(ctx, hipEventCreateWithFlags(&ctx->ref_stage_done[k], hipEventDisableTiming));
    }
    ftk_ctx::RefImage r;
    r.bytes = n_bytes;
    r.kind = kind;
    size_t cap = 0;
    if ((rc = ref_block_take(ctx, (size_t)n_bytes + 32, &r.d, &cap))) return rc;
    r.cap = (int64_t)cap;
    // (a chunk is only rewritten once the copy that last read it has finished - the event of a PREVIOUS call's last
    // chunks included: the call returns with its final copies still in flight; an event never recorded is complete)
    hipError_t e = hipMemsetAsync((char*)r.d + n_bytes, 0, 32, ctx->stream);
    int k = 0;
    // the file (page cache) -> a page-locked chunk on four pread threads -> the device, the next chunk read while the
    // previous one is on its way
    for (int64_t off = 0; off < n_bytes && e == hipSuccess; off += (int64_t)kRefStageBytes, k ^= 1) {
        const size_t n = (size_t)std::min<int64_t>((int64_t)kRefStageBytes, n_bytes - off);
        e = hipEventSynchronize(ctx->ref_stage_done[k]);
        if (e != hipSuccess) break;
        std::atomic<int> bad{0};
        const int nt = n >= (size_t(4) << 20) ? 4 : 1;
        char* dst = (char*)ctx->ref_stage[k];
        ftk_host::parallel_run(nt, [&](int t) {
            size_t a = n * (size_t)t / nt;
            const size_t b = n * (size_t)(t + 1) / nt;
            while (a < b) {
                const ssize_t got = pread(fd, dst + a, b - a, (off_t)(file_offset + off + (int64_t)a));
                if (got <= 0) { bad.store(1); return; }
                a += (size_t)got;
            }
        });
        if (bad.load()) {
            (void)hipStreamSynchronize(ctx->stream);
            ref_block_give(ctx, r.d, cap);
            return fail(ctx, FTK_ERR_IO, "%s is shorter than the reference image it should hold", path);
        }
        e = hipMemcpyAsync((char*)r.d + off, dst, n, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipEventRecord(ctx->ref_stage_done[k], ctx->stream);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamSynchronize(ctx->stream);
        ref_block_give(ctx, r.d, cap);
        return fail(ctx, FTK_ERR_HIP, "reference upload failed: %s", hipGetErrorString(e));
    }
    ctx->refs[ref_id] = r;  // (kernels of this stream run behind the copies; the staging chunks wait on their events)
    return FTK_OK;
}

int ftk_ref_release(ftk_ctx* ctx, int ref_id) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    if (ctx->refs.find(ref_id) == ctx->refs.end()) return fail(ctx, FTK_ERR_NO_CONTIG, "reference image %d is not loaded", ref_id);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    return ref_drop(ctx, ref_id);
}

int ftk_ref_set_layout(ftk_ctx* ctx, int ref_id, int64_t chrom_len, int32_t line_bases, int32_t line_width,
                       const int32_t* nblock_start, const int32_t* nblock_end, int64_t n_nblocks) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    auto it = ctx->refs.find(ref_id);
    if (it == ctx->refs.end()) return fail(ctx, FTK_ERR_NO_CONTIG, "reference image %d is not loaded", ref_id);
    ftk_ctx::RefImage& r = it->second;
    if (chrom_len < 0 || chrom_len > INT32_MAX) return fail(ctx, FTK_ERR_INVALID, "chrom_len out of range");
    if (n_nblocks < 0 || n_nblocks > (1 << 28) || (n_nblocks > 0 && (!nblock_start || !nblock_end)))
        return fail(ctx, FTK_ERR_INVALID, "bad N-block arguments");
    if (r.kind == FTK_REF_FASTA_TEXT) {
        if (chrom_len > 0 && (line_bases <= 0 || line_width < line_bases))
            return fail(ctx, FTK_ERR_INVALID, "FASTA images need line_bases > 0 and line_width >= line_bases");
        if (chrom_len > 0 && (chrom_len / line_bases) * line_width + chrom_len % line_bases > r.bytes)
            return fail(ctx, FTK_ERR_INVALID, "chrom_len does not fit the uploaded FASTA text");
    } else if ((chrom_len + 3) / 4 > r.bytes) {
        return fail(ctx, FTK_ERR_INVALID, "chrom_len does not fit the uploaded 2bit image");
    }
    for (int64_t i = 0; i < n_nblocks; ++i)
        if (nblock_start[i] < 0 || nblock_end[i] < nblock_start[i] || (i && nblock_start[i] < nblock_end[i - 1]))
            return fail(ctx, FTK_ERR_INVALID, "N blocks must be sorted and disjoint");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (r.d_nblk) (void)hipFree(r.d_nblk);
    r.d_nblk = nullptr;
    r.n_nblk = 0;
    if (n_nblocks) {
        HIPCHK(ctx, hipMalloc((void**)&r.d_nblk, (size_t)n_nblocks * 8));
        HIPCHK(ctx, hipMemcpy(r.d_nblk, nblock_start, (size_t)n_nblocks * 4, hipMemcpyHostToDevice));
        HIPCHK(ctx, hipMemcpy(r.d_nblk + n_nblocks, nblock_end, (size_t)n_nblocks * 4, hipMemcpyHostToDevice));
        r.n_nblk = (int32_t)n_nblocks;
    }
    r.chrom_len = chrom_len;
    r.line_bases = line_bases;
    r.line_width = line_width;
    return FTK_OK;
}

int ftk_motif_counts(ftk_ctx* ctx, int contig_id, int ref_id, const int32_t* w_start, const int32_t* w_end,
                     int64_t n_win, const ftk_motif* motif, int32_t mapq_min, int32_t fetch_mode,
                     uint32_t* counts_out, int64_t* nfrag_out, int64_t* err_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    auto it = ctx->refs.find(ref_id);
    if (it == ctx->refs.end()) return fail(ctx, FTK_ERR_NO_CONTIG, "reference image %d is not loaded", ref_id);
    const ftk_ctx::RefImage& ri = it->second;
    if (ri.chrom_len < 0) return fail(ctx, FTK_ERR_INVALID, "reference image %d has no layout (ftk_ref_set_layout)", ref_id);
    if (!motif) return fail(ctx, FTK_ERR_INVALID, "motif is NULL");
    if (motif->k < 1 || motif->k > 7) return fail(ctx, FTK_ERR_INVALID, "k must be in [1, 7]");
    if (n_win > 0 && (!counts_out || !err_out)) return fail(ctx, FTK_ERR_INVALID, "NULL output pointer");
    MotifParams mp{};
    mp.img = (const uint8_t*)ri.d;
    mp.nblk_start = ri.d_nblk;
    mp.nblk_end = ri.d_nblk ? ri.d_nblk + ri.n_nblk : nullptr;
    mp.n_nblk = ri.n_nblk;
    mp.kind = ri.kind;
    mp.chrom_len = (int)ri.chrom_len;
    mp.line_bases = ri.line_bases > 0 ? ri.line_bases : 1;
    mp.line_width = ri.line_width > 0 ? ri.line_width : 1;
    mp.k = motif->k;
    mp.f_off = motif->fwd_offset;
    mp.r_off = motif->rev_offset;
    mp.both = motif->both_strands != 0;
    mp.neg = motif->negative_strand != 0;
    mp.guard = motif->guard;
    mp.rev_err = motif->rev_oob_is_error != 0;
    ftk_filter f{mapq_min, FTK_LEN_OPEN, FTK_LEN_OPEN, FTK_POLICY_ANY, fetch_mode};
    FeatCall fc;
    fc.f = &f;
    fc.count_out = nfrag_out;
    fc.hist_out = counts_out;
    fc.overflow_out = err_out;
    fc.len_lo = 0;
    fc.n_bins = 1 << (2 * motif->k);
    fc.motif = &mp;
    return features_common(ctx, contig_id, w_start, w_end, n_win, fc);
}

int ftk_ref_gc_counts(ftk_ctx* ctx, int ref_id, const int64_t* range_lo, const int64_t* range_hi, int64_t n,
                      int64_t* gc_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    auto it = ctx->refs.find(ref_id);
    if (it == ctx->refs.end()) return fail(ctx, FTK_ERR_NO_CONTIG, "reference image %d is not loaded", ref_id);
    if (n < 0 || n > INT32_MAX) return fail(ctx, FTK_ERR_INVALID, "n out of range");
    if (n == 0) return FTK_OK;
    if (!range_lo || !range_hi || !gc_out) return fail(ctx, FTK_ERR_INVALID, "NULL argument");
    const ftk_ctx::RefImage& r = it->second;
    const int64_t limit = r.kind == FTK_REF_2BIT ? r.bytes * 4 : r.bytes;
    if (!is_device_ptr(range_lo))
        for (int64_t i = 0; i < n; ++i)
            if (range_lo[i] < 0 || range_hi[i] > limit)
                return fail(ctx, FTK_ERR_INVALID, "range %lld outside the reference image", (long long)i);
    HIPCHK(ctx, hipSetDevice(ctx->device));
    const bool out_dev = is_device_ptr(gc_out);
    int rc = reserve_scratch(ctx, 3 * align_up(n * 8));
    if (rc) return rc;
    Arena a(ctx);
    int64_t* b_lo = a.take<int64_t>(n);
    int64_t* b_hi = a.take<int64_t>(n);
    int64_t* d_out = out_dev ? gc_out : a.take<int64_t>(n);
    const int64_t *d_lo, *d_hi;
    if ((rc = stage_in(ctx, range_lo, n, b_lo, &d_lo))) return rc;
    if ((rc = stage_in(ctx, range_hi, n, b_hi, &d_hi))) return rc;
    launch_gc_count(ctx->stream, (const uint8_t*)r.d, r.bytes, r.kind, d_lo, d_hi, (int)n, d_out);
    HIPCHK(ctx, hipGetLastError());
    if (!out_dev) {
        HIPCHK(ctx, hipMemcpyAsync(gc_out, d_out, n * 8, hipMemcpyDeviceToHost, ctx->stream));
        HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    }
    return FTK_OK;
}

// ---- DEFLATE on the device (csrc/ftk_inflate.hip) -------------------------------------------------------
int ftk_bgzf_inflate_device(ftk_ctx* ctx, const uint8_t* file_bytes, int64_t n, uint8_t* out, int64_t cap, int64_t* n_out) {
    if (!ctx) return fail(nullptr, FTK_ERR_INVALID, "ctx is NULL");
    if (!file_bytes || n < 0 || (cap > 0 && !out) || !n_out) return fail(ctx, FTK_ERR_INVALID, "bad arguments");
    if (n >= (int64_t(1) << 32) - 65536) return fail(ctx, FTK_ERR_INVALID, "at most 4 GB per call");
    // walk the BGZF blocks: 12-byte gzip header with FEXTRA, the BC subfield gives the block size
    std::vector<ftk::InflateBlock> tab;
    std::vector<uint32_t> want_crc;
    uint64_t off = 0, total = 0;
    while (off < (uint64_t)n) {
        const uint8_t* p = file_bytes + off;
        if (off + 18 > (uint64_t)n || p[0] != 31 || p[1] != 139 || p[2] != 8 || !(p[3] & 4))
            return fail(ctx, FTK_ERR_FORMAT, "not a BGZF block at byte %llu", (unsigned long long)off);
        const unsigned xlen = p[10] | (p[11] << 8);
        unsigned bsize = 0;
        for (unsigned x = 0; x + 4 <= xlen && off + 12 + x + 4 <= (uint64_t)n;) {
            const uint8_t* f = p + 12 + x;
            const unsigned slen = f[2] | (f[3] << 8);
            if (f[0] == 66 && f[1] == 67 && slen == 2 && off + 12 + x + 6 <= (uint64_t)n) bsize = (f[4] | (f[5] << 8)) + 1u;
            x += 4 + slen;
        }
        const uint64_t q = off + 12 + xlen;
        if (!bsize || off + bsize > (uint64_t)n || q + 8 > off + bsize)
            return fail(ctx, FTK_ERR_FORMAT, "corrupt BGZF block at byte %llu", (unsigned long long)off);
        const uint8_t* tr = p + bsize - 8;
        const uint32_t isize = (uint32_t)tr[4] | ((uint32_t)tr[5] << 8) | ((uint32_t)tr[6] << 16) | ((uint32_t)tr[7] << 24);
        if (isize > 65536u || total + isize >= (uint64_t(1) << 32)) return fail(ctx, FTK_ERR_FORMAT, "BGZF block too large");
        tab.push_back({(uint32_t)q, (uint32_t)(off + bsize - 8 - q), (uint32_t)total, isize});
        want_crc.push_back((uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24));
        total += isize;
        off += bsize;
    }
    *n_out = (int64_t)total;
    if ((int64_t)total > cap) return fail(ctx, FTK_ERR_INVALID, "output holds %lld bytes, %lld needed", (long long)cap, (long long)total);
    if (tab.empty() || total == 0) return FTK_OK;
    HIPCHK(ctx, hipSetDevice(ctx->device));
    int rc = reserve_scratch(ctx, align_up((size_t)n + 8) + align_up(tab.size() * sizeof(tab[0])) + align_up(tab.size() * 4) +
                                      align_up((size_t)total + 8192) + 4096);
    if (rc) return rc;
    Arena a(ctx);
    uint8_t* d_comp = a.take<uint8_t>((size_t)n + 8);
    ftk::InflateBlock* d_tab = a.take<ftk::InflateBlock>(tab.size());
    ftk::InflateStatus* d_st = a.take<ftk::InflateStatus>(1);
    uint32_t* d_crc = a.take<uint32_t>(tab.size());
    uint8_t* d_out = a.take<uint8_t>((size_t)total + 4096 + 16);
    d_out = (uint8_t*)(((uintptr_t)d_out + 4095) & ~(uintptr_t)4095);
    // The kernel addresses its input by 32-BIT BIT POSITIONS from the pointer it is given: a launch takes blocks whose
    // compressed bytes lie within 2^28 bytes of its first one (the streams' pieces are 48-96 MB; an image of more than
    // that goes in several launches, each from a base of its own - block offsets are rebased here, before the table goes up).
    struct Run {
        size_t first, count, base;
    };
    std::vector<Run> runs;
    for (size_t i = 0; i < tab.size();) {
        const size_t base = (size_t)tab[i].in_off & ~(size_t)3;
        size_t j = i;
        while (j < tab.size() && (size_t)tab[j].in_off + tab[j].in_len - base < (size_t(1) << 28)) ++j;
        if (j == i) return fail(ctx, FTK_ERR_FORMAT, "BGZF block too large");  // (cannot happen: a block is at most 64 KB)
        runs.push_back({i, j - i, base});
        for (size_t k = i; k < j; ++k) tab[k].in_off -= (uint32_t)base;
        i = j;
    }
    HIPCHK(ctx, hipMemcpyAsync(d_comp, file_bytes, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(d_tab, tab.data(), tab.size() * sizeof(tab[0]), hipMemcpyHostToDevice, ctx->stream));
    // (FTK_INFLATE_VECTOR_MATCHES=1: the launch shape BAM streams use - a window's matches resolved on the lanes side
    // by side; read per call so that the tests can hold both shapes against zlib)
    const char* vm = getenv("FTK_INFLATE_VECTOR_MATCHES");
    ftk::InflateStatus st{};
    for (size_t r = 0; r < runs.size(); ++r) {
        HIPCHK(ctx, hipMemsetAsync(d_st, 0, sizeof(*d_st), ctx->stream));
        ftk::inflate_launch(ctx->stream, d_comp + runs[r].base, d_tab + runs[r].first, (int)runs[r].count, d_out, d_st,
                            d_crc + runs[r].first, vm && atoi(vm) != 0);
        HIPCHK(ctx, hipGetLastError());
        if (r + 1 < runs.size()) {  // (the status words are the launch's own: read before the next one clears them)
            ftk::InflateStatus sr{};
            HIPCHK(ctx, hipMemcpyAsync(&sr, d_st, sizeof(sr), hipMemcpyDeviceToHost, ctx->stream));
            HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
            if (sr.n_bad)
                return fail(ctx, FTK_ERR_FORMAT, "device inflate: %u of %zu BGZF blocks did not decode (block %zu: reason %u)", sr.n_bad,
                            tab.size(), runs[r].first + sr.first_bad, sr.reason);
        }
    }
    const size_t last_first = runs.back().first;
    std::vector<uint32_t> got_crc(tab.size());
    HIPCHK(ctx, hipMemcpyAsync(&st, d_st, sizeof(st), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(got_crc.data(), d_crc, tab.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipMemcpyAsync(out, d_out, (size_t)total, hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(ctx, hipStreamSynchronize(ctx->stream));
    if (st.n_bad)
        return fail(ctx, FTK_ERR_FORMAT, "device inflate: %u of %zu BGZF blocks did not decode (block %zu: reason %u)", st.n_bad,
                    tab.size(), last_first + st.first_bad, st.reason);
    for (size_t k = 0; k < tab.size(); ++k)  // the gzip trailer's CRC-32 against the one computed on the device
        if (got_crc[k] != want_crc[k])
            return fail(ctx, FTK_ERR_FORMAT, "device inflate: CRC mismatch in BGZF block %zu (%08x, trailer says %08x)", k,
                        got_crc[k], want_crc[k]);
    return FTK_OK;
}

}  // extern "C"

// ---- the exchange between the ranks of one node: RCCL behind the C ABI ---------------------------------------------
// SURVEY section 8-b's export list ends with the two collectives that stand where the reference gathers its Pool's
// results (frag/_delfi.py:289-300: imap over the bins -> one list; frag/_coverage.py:215-227: the interval counts and
// the genome-wide total): one all-gather of fixed-size int64 rows and one int64 all-reduce.  One rank per GPU; the
// library is NOT linked against RCCL: librccl is resolved when the first communicator is created (a process that holds
// torch's copy already re-uses it: same soname), so one-GPU hosts never load it.
// A communicator belongs to a ctx.  A collective is enqueued on the communicator's OWN stream behind whatever the ctx
// stream holds at the call (an event), so kernels launched on the ctx stream afterwards run beside it; device results
// are complete for the ctx stream after ftk_comm_join (another event, no host wait), host results when the call returns.
#include <rccl/rccl.h>

#include <sys/stat.h>

struct ftk_comm {
    ftk_ctx* ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    hipStream_t stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_done = nullptr;
    void* stage = nullptr;  // device staging for host buffers
    size_t stage_bytes = 0;
    std::string id_path;    // rank 0: the rendezvous file it wrote (removed with the communicator)
};

namespace {

struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi* rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // a librccl already in the process (torch's) is re-used: same soname
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) return;
        api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
        api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
        api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
        api.Send = (decltype(api.Send))dlsym(api.lib, "ncclSend");
        api.Recv = (decltype(api.Recv))dlsym(api.lib, "ncclRecv");
        api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
        api.CommAbort = (decltype(api.CommAbort))dlsym(api.lib, "ncclCommAbort");
        api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    });
    const bool ok = api.lib && api.GetUniqueId && api.CommInitRank && api.AllGather && api.AllReduce && api.Send &&
                    api.Recv && api.CommDestroy;
    return ok ? &api : nullptr;
}

#define RCCLCHK(ctx, call)                                                                                        \
    do {                                                                                                          \
        ncclResult_t r_ = (call);                                                                                 \
        if (r_ != ncclSuccess)                                                                                    \
            return fail(ctx, FTK_ERR_HIP, "%s: %s", #call, rccl()->GetErrorString ? rccl()->GetErrorString(r_) : "RCCL error"); \
    } while (0)

constexpr size_t kIdHex = 2 * NCCL_UNIQUE_ID_BYTES;

void id_to_hex(const ncclUniqueId& id, char* hex) {
    static const char d[] = "0123456789abcdef";
    for (int i = 0; i < NCCL_UNIQUE_ID_BYTES; ++i) {
        hex[2 * i] = d[((unsigned char)id.internal[i]) >> 4];
        hex[2 * i + 1] = d[((unsigned char)id.internal[i]) & 15];
    }
    hex[kIdHex] = 0;
}

bool hex_to_id(const char* hex, ncclUniqueId* id) {
    auto v = [](char c) { return c >= '0' && c <= '9' ? c - '0' : c >= 'a' && c <= 'f' ? c - 'a' + 10 : c >= 'A' && c <= 'F' ? c - 'A' + 10 : -1; };
    for (int i = 0; i < NCCL_UNIQUE_ID_BYTES; ++i) {
        const int a = v(hex[2 * i]), b = a < 0 ? -1 : v(hex[2 * i + 1]);
        if (a < 0 || b < 0) return false;
        id->internal[i] = (char)(a << 4 | b);
    }
    return true;
}

bool looks_like_hex_id(const char* s) {
    if (strlen(s) != kIdHex) return false;
    ncclUniqueId t;
    return hex_to_id(s, &t);
}

// device staging of the communicator (host buffers of a collective), grown on demand
int comm_stage(ftk_comm* c, size_t bytes) {
    if (bytes <= c->stage_bytes) return FTK_OK;
    ftk_ctx* ctx = c->ctx;
    HIPCHK(ctx, hipStreamSynchronize(c->stream));
    if (c->stage) (void)hipFree(c->stage);
    c->stage = nullptr;
    c->stage_bytes = 0;
    HIPCHK(ctx, hipMalloc(&c->stage, align_up(bytes, 1 << 16)));
    c->stage_bytes = align_up(bytes, 1 << 16);
    return FTK_OK;
}

// the communicator's stream behind the ctx stream's work of this moment
int comm_fork(ftk_comm* c) {
    HIPCHK(c->ctx, hipSetDevice(c->ctx->device));
    HIPCHK(c->ctx, hipEventRecord(c->ev_fork, c->ctx->stream));
    HIPCHK(c->ctx, hipStreamWaitEvent(c->stream, c->ev_fork, 0));
    return FTK_OK;
}

}  // namespace

// RCCL announces itself on STDOUT when a communicator comes up ("RCCL version : ...", five lines) - in the way of a host
// whose stdout is its result (bench.py's one JSON line).  While the library is inside RCCL's set-up calls, file descriptor
// 1 points where 2 points; FTK_COMM_BANNER=1 leaves it alone.
struct QuietStdout {
    int saved = -1;
    QuietStdout() {
        static const bool keep = getenv("FTK_COMM_BANNER") && atoi(getenv("FTK_COMM_BANNER")) != 0;
        if (keep) return;
        fflush(stdout);
        saved = dup(1);
        if (saved >= 0 && dup2(2, 1) < 0) {
            close(saved);
            saved = -1;
        }
    }
    ~QuietStdout() {
        if (saved < 0) return;
        fflush(stdout);
        (void)dup2(saved, 1);
        close(saved);
    }
};

extern "C" {

int ftk_comm_unique_id(char* hex_out) {
    if (!hex_out) return fail(nullptr, FTK_ERR_INVALID, "hex_out is NULL");
    QuietStdout quiet;
    RcclApi* a = rccl();
    if (!a) return fail(nullptr, FTK_ERR_NO_DEVICE, "librccl could not be loaded");
    ncclUniqueId id;
    RCCLCHK(nullptr, a->GetUniqueId(&id));
    id_to_hex(id, hex_out);
    return FTK_OK;
}

// The rendezvous file.  Content: the 256 hex digits of the id, then - when the launcher gave the job a nonce
// (FTK_COMM_NONCE: sharding.launch_ranks makes one per launch; comm.py passes TORCHELASTIC_RUN_ID on) - a newline and
// the nonce.  A reader takes a file only if (i) its nonce is the reader's own, when the reader has one, and (ii) it was
// written after this library was loaded into the reader (minus a slack for ranks that start a moment apart): the file
// a killed job left behind - rank 0 removes it only once its communicator is up - is older than that and is ignored
// instead of being joined, which would hang in ncclCommInitRank for ever.
static const time_t g_loaded_at = time(nullptr);

static const char* comm_nonce() {
    const char* e = getenv("FTK_COMM_NONCE");
    return e ? e : "";
}

int ftk_comm_create(ftk_ctx* ctx, int rank, int world, const char* id_hex_or_path, ftk_comm** out) {
    if (!ctx || !out) return fail(ctx, FTK_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(ctx, FTK_ERR_INVALID, "bad rank %d / world %d", rank, world);
    if (!id_hex_or_path && world > 1) return fail(ctx, FTK_ERR_INVALID, "ranks of a job need a common id (hex digits or a file path)");
    RcclApi* a = rccl();
    if (!a) return fail(ctx, FTK_ERR_NO_DEVICE, "librccl could not be loaded (dlopen librccl.so.1)");
    HIPCHK(ctx, hipSetDevice(ctx->device));
    static const double limit_s = getenv("FTK_COMM_TIMEOUT_S") ? atof(getenv("FTK_COMM_TIMEOUT_S")) : 600.0;
    ncclUniqueId uid;
    std::string wrote;
    auto fresh_id = [&]() -> ncclResult_t {
        QuietStdout quiet;  // (only around the RCCL calls that print the banner: other threads' stdout stays theirs)
        return a->GetUniqueId(&uid);
    };
    if (!id_hex_or_path) {
        RCCLCHK(ctx, fresh_id());
    } else if (looks_like_hex_id(id_hex_or_path)) {
        hex_to_id(id_hex_or_path, &uid);
    } else if (rank == 0) {
        // rendezvous through a file: rank 0 writes the id next to the final name - a new file of its own (O_EXCL), never
        // through a link somebody planted (O_NOFOLLOW), readable by its user alone - and renames it into place, so that
        // a reader never sees part of it.  A file already at the final name is a dead job's: it goes first.
        (void)unlink(id_hex_or_path);
        RCCLCHK(ctx, fresh_id());
        char hex[kIdHex + 1];
        id_to_hex(uid, hex);
        const std::string body = std::string(hex) + "\n" + std::string(comm_nonce());
        const std::string tmp = std::string(id_hex_or_path) + ".tmp" + std::to_string((long long)getpid());
        (void)unlink(tmp.c_str());
        const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW | O_CLOEXEC, 0600);
        bool ok = fd >= 0 && write(fd, body.data(), body.size()) == (ssize_t)body.size();
        if (fd >= 0) ok = (close(fd) == 0) && ok;
        if (!ok || rename(tmp.c_str(), id_hex_or_path) != 0) {
            const int err = errno;
            if (fd >= 0) (void)unlink(tmp.c_str());
            return fail(ctx, FTK_ERR_IO, "cannot write the rendezvous file %s: %s", id_hex_or_path, strerror(err));
        }
        wrote = id_hex_or_path;
    } else {
        const auto t0 = std::chrono::steady_clock::now();
        const std::string want_nonce(comm_nonce());
        for (;;) {
            const int fd = open(id_hex_or_path, O_RDONLY | O_NOFOLLOW | O_CLOEXEC);
            if (fd >= 0) {
                char buf[kIdHex + 258] = {0};
                struct stat st;
                const bool have = fstat(fd, &st) == 0;
                const ssize_t got = read(fd, buf, sizeof(buf) - 1);
                close(fd);
                if (have && got >= (ssize_t)kIdHex + 1 && buf[kIdHex] == '\n') {
                    buf[got] = 0;
                    buf[kIdHex] = 0;
                    const bool nonce_ok = want_nonce.empty() || want_nonce == (buf + kIdHex + 1);
                    const bool recent = st.st_mtime + 10 >= g_loaded_at;  // (written by THIS launch, not left by an earlier one)
                    if (nonce_ok && recent && hex_to_id(buf, &uid)) break;
                }
            }
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit_s)
                return fail(ctx, FTK_ERR_IO, "rank %d: no communicator id of this launch in %s after %.0f s (did rank 0 start? do all ranks "
                            "share FTK_COMM_ID_FILE / FTK_COMM_NONCE? a file older than this process is ignored)", rank, id_hex_or_path, limit_s);
            usleep(2000);
        }
    }
    ftk_comm* c = new (std::nothrow) ftk_comm();
    if (!c) return fail(ctx, FTK_ERR_OOM, "out of host memory");
    c->ctx = ctx;
    c->rank = rank;
    c->world = world;
    c->id_path = wrote;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        ftk_comm_destroy(c);
        return fail(ctx, FTK_ERR_HIP, "communicator setup failed: %s", hipGetErrorString(e));
    }
    // ncclCommInitRank has no timeout of its own and waits for ever for a rank that never comes (one that died, or read
    // another launch's id): it runs on a helper thread, and a launch whose ranks do not meet within the limit fails here
    // with a message instead of hanging (the helper is abandoned; the process is expected to exit on this error)
    auto init = std::make_shared<std::promise<ncclResult_t>>();
    std::future<ncclResult_t> done = init->get_future();
    {
        ncclComm_t* dst = &c->comm;
        const int dev = ctx->device;
        std::thread([a, dst, world, uid, rank, dev, init] {
            (void)hipSetDevice(dev);
            QuietStdout quiet;
            init->set_value(a->CommInitRank(dst, world, uid, rank));
        }).detach();
    }
    if (done.wait_for(std::chrono::duration<double>(limit_s)) != std::future_status::ready) {
        // (c is leaked on purpose: the helper may still write c->comm)
        return fail(ctx, FTK_ERR_IO, "rank %d of %d: the ranks did not meet in ncclCommInitRank within %.0f s (FTK_COMM_TIMEOUT_S); rendezvous %s",
                    rank, world, limit_s, id_hex_or_path ? (looks_like_hex_id(id_hex_or_path) ? "by id" : id_hex_or_path) : "none");
    }
    const ncclResult_t r = done.get();
    if (r != ncclSuccess) {
        c->comm = nullptr;
        ftk_comm_destroy(c);
        return fail(ctx, FTK_ERR_HIP, "ncclCommInitRank(rank %d of %d): %s", rank, world, a->GetErrorString ? a->GetErrorString(r) : "RCCL error");
    }
    // (every rank has read the id by now - the initialisation is collective - so the rendezvous file goes at once: a job
    // that dies later leaves nothing for the next one to trip over)
    if (!c->id_path.empty()) {
        (void)remove(c->id_path.c_str());
        c->id_path.clear();
    }
    *out = c;
    return FTK_OK;
}

int ftk_comm_size(const ftk_comm* comm, int* rank_out, int* world_out) {
    if (!comm) return fail(nullptr, FTK_ERR_INVALID, "comm is NULL");
    if (rank_out) *rank_out = comm->rank;
    if (world_out) *world_out = comm->world;
    return FTK_OK;
}

int ftk_comm_join(ftk_comm* comm) {
    if (!comm) return fail(nullptr, FTK_ERR_INVALID, "comm is NULL");
    HIPCHK(comm->ctx, hipSetDevice(comm->ctx->device));
    HIPCHK(comm->ctx, hipStreamWaitEvent(comm->ctx->stream, comm->ev_done, 0));
    return FTK_OK;
}

int ftk_allgather_i64(ftk_comm* comm, const int64_t* send, int64_t n, int64_t* recv) {
    if (!comm) return fail(nullptr, FTK_ERR_INVALID, "comm is NULL");
    ftk_ctx* ctx = comm->ctx;
    if (n < 0 || (n > 0 && (!send || !recv))) return fail(ctx, FTK_ERR_INVALID, "bad arguments");
    if (n == 0) return FTK_OK;
    int rc = comm_fork(comm);
    if (rc) return rc;
    const bool s_dev = is_device_ptr(send), r_dev = is_device_ptr(recv);
    const size_t b = (size_t)n * 8, b_all = b * (size_t)comm->world;
    const size_t s_off = 0, r_off = s_dev ? 0 : align_up(b);
    if ((rc = comm_stage(comm, (s_dev ? 0 : align_up(b)) + (r_dev ? 0 : align_up(b_all))))) return rc;
    const int64_t* d_send = send;
    if (!s_dev) {
        HIPCHK(ctx, hipMemcpyAsync((char*)comm->stage + s_off, send, b, hipMemcpyHostToDevice, comm->stream));
        d_send = (const int64_t*)((char*)comm->stage + s_off);
    }
    int64_t* d_recv = r_dev ? recv : (int64_t*)((char*)comm->stage + r_off);
    RCCLCHK(ctx, rccl()->AllGather(d_send, d_recv, (size_t)n, ncclInt64, comm->comm, comm->stream));
    if (!r_dev) HIPCHK(ctx, hipMemcpyAsync(recv, d_recv, b_all, hipMemcpyDeviceToHost, comm->stream));
    HIPCHK(ctx, hipEventRecord(comm->ev_done, comm->stream));
    if (!r_dev || !s_dev) HIPCHK(ctx, hipStreamSynchronize(comm->stream));  // host buffers: done when the call returns
    return FTK_OK;
}

int ftk_allreduce_sum_i64(ftk_comm* comm, int64_t* values, int64_t n) {
    if (!comm) return fail(nullptr, FTK_ERR_INVALID, "comm is NULL");
    ftk_ctx* ctx = comm->ctx;
    if (n < 0 || (n > 0 && !values)) return fail(ctx, FTK_ERR_INVALID, "bad arguments");
    if (n == 0) return FTK_OK;
    int rc = comm_fork(comm);
    if (rc) return rc;
    const bool dev = is_device_ptr(values);
    const size_t b = (size_t)n * 8;
    int64_t* d = values;
    if (!dev) {
        if ((rc = comm_stage(comm, b))) return rc;
        d = (int64_t*)comm->stage;
        HIPCHK(ctx, hipMemcpyAsync(d, values, b, hipMemcpyHostToDevice, comm->stream));
    }
    RCCLCHK(ctx, rccl()->AllReduce(d, d, (size_t)n, ncclInt64, ncclSum, comm->comm, comm->stream));
    if (!dev) HIPCHK(ctx, hipMemcpyAsync(values, d, b, hipMemcpyDeviceToHost, comm->stream));
    HIPCHK(ctx, hipEventRecord(comm->ev_done, comm->stream));
    if (!dev) HIPCHK(ctx, hipStreamSynchronize(comm->stream));
    return FTK_OK;
}

// point to point, for the compressed output sections a rank hands to the writing rank (frag/_multi_wps.py:300-341 runs
// its writer in the parent of the Pool): bytes in chunks of at most 1 GiB through the staging block
static int comm_p2p(ftk_comm* comm, int peer, void* data, int64_t n_bytes, bool sending) {
    if (!comm) return fail(nullptr, FTK_ERR_INVALID, "comm is NULL");
    ftk_ctx* ctx = comm->ctx;
    if (peer < 0 || peer >= comm->world || peer == comm->rank || n_bytes < 0 || (n_bytes > 0 && !data))
        return fail(ctx, FTK_ERR_INVALID, "bad arguments (peer %d)", peer);
    if (n_bytes == 0) return FTK_OK;
    int rc = comm_fork(comm);
    if (rc) return rc;
    const bool dev = is_device_ptr(data);
    constexpr size_t kChunk = size_t(1) << 28;
    if (!dev && (rc = comm_stage(comm, std::min<size_t>((size_t)n_bytes, kChunk)))) return rc;
    for (size_t off = 0; off < (size_t)n_bytes; off += kChunk) {
        const size_t len = std::min(kChunk, (size_t)n_bytes - off);
        char* h = (char*)data + off;
        void* d = dev ? (void*)h : comm->stage;
        if (sending) {
            if (!dev) HIPCHK(ctx, hipMemcpyAsync(d, h, len, hipMemcpyHostToDevice, comm->stream));
            RCCLCHK(ctx, rccl()->Send(d, len, ncclUint8, peer, comm->comm, comm->stream));
        } else {
            RCCLCHK(ctx, rccl()->Recv(d, len, ncclUint8, peer, comm->comm, comm->stream));
            if (!dev) HIPCHK(ctx, hipMemcpyAsync(h, d, len, hipMemcpyDeviceToHost, comm->stream));
        }
        if (!dev) HIPCHK(ctx, hipStreamSynchronize(comm->stream));  // the staging block is reused by the next chunk
    }
    HIPCHK(ctx, hipEventRecord(comm->ev_done, comm->stream));
    return FTK_OK;
}

int ftk_comm_send(ftk_comm* comm, int dst, const void* data, int64_t n_bytes) {
    return comm_p2p(comm, dst, const_cast<void*>(data), n_bytes, true);
}

int ftk_comm_recv(ftk_comm* comm, int src, void* data, int64_t n_bytes) { return comm_p2p(comm, src, data, n_bytes, false); }

void ftk_comm_destroy(ftk_comm* comm) {
    if (!comm) return;
    if (comm->ctx) (void)hipSetDevice(comm->ctx->device);
    if (comm->stream) (void)hipStreamSynchronize(comm->stream);
    if (comm->comm && rccl()) (void)rccl()->CommDestroy(comm->comm);
    if (comm->stage) (void)hipFree(comm->stage);
    if (comm->ev_fork) (void)hipEventDestroy(comm->ev_fork);
    if (comm->ev_done) (void)hipEventDestroy(comm->ev_done);
    if (comm->stream) (void)hipStreamDestroy(comm->stream);
    if (!comm->id_path.empty()) (void)remove(comm->id_path.c_str());
    delete comm;
}

}  // extern "C"
