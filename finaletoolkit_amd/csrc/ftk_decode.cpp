// Host-side decoders: BGZF/gzip fragment text (frag.gz, BED6 bed.gz) and BAM
// -> per-contig SoA columns.  Pure host code (zlib + std::thread): usable and
// tested without a GPU.  Follows io/alignment.py:270-302 (_fetch_tabix) and
// io/alignment.py:60-71,242-268 (_fetch_sam) of the reference; mapq is kept as
// a column (the cut is applied by the kernels).
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <future>
#include <memory>
#include <mutex>
#include <numeric>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include <dlfcn.h>
#include <fcntl.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>

#include "ftk.h"
#include "ftk_host.h"
#include "ftk_inflate.h"
#include "ftk_bamparse.h"
#include "ftk_bamrule.h"
#include "ftk_textparse.h"

namespace {

thread_local std::string g_decode_err;

int dfail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_decode_err = buf;
    return code;
}

// Persistent worker threads for the decoders' parallel regions.  A decode of one file opens a few
// dozen short regions (inflate / parse / pack per piece); creating up to 256 threads for each of them
// cost more than the work itself.  Regions are serialised (one runs at a time); workers are created on
// demand and live for the rest of the process (a forked child starts with a fresh pool).
class WorkPool {
    std::mutex mu, region;
    std::condition_variable cv_work, cv_done;
    std::vector<std::thread> threads;
    const std::function<void(int)>* fn = nullptr;
    int want = 0, pending = 0;
    unsigned long long gen = 0;

    void worker(int idx, unsigned long long seen) {
        std::unique_lock<std::mutex> lk(mu);
        for (;;) {
            cv_work.wait(lk, [&] { return gen != seen; });
            seen = gen;
            if (idx <= want) {
                const std::function<void(int)>* f = fn;
                lk.unlock();
                (*f)(idx);
                lk.lock();
                if (--pending == 0) cv_done.notify_all();
            }
        }
    }

public:
    // fn(0) runs on the caller, fn(1..n-1) on pool threads; returns when all are done
    void run(int n, const std::function<void(int)>& f) {
        if (n <= 1) { f(0); return; }
        std::lock_guard<std::mutex> one(region);
        {
            std::unique_lock<std::mutex> lk(mu);
            while ((int)threads.size() < n - 1) {
                const int idx = (int)threads.size() + 1;
                const unsigned long long g = gen;
                threads.emplace_back([this, idx, g] { worker(idx, g); });
                threads.back().detach();
            }
            fn = &f;
            want = n - 1;
            pending = n - 1;
            ++gen;
        }
        cv_work.notify_all();
        f(0);
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return pending == 0; });
    }
};

WorkPool* g_pool = nullptr;
std::once_flag g_pool_fork_once;

void parallel_run(int n, const std::function<void(int)>& f) {
    static std::mutex init;
    WorkPool* p;
    {
        std::lock_guard<std::mutex> lk(init);
        std::call_once(g_pool_fork_once, [] { pthread_atfork(nullptr, nullptr, [] { g_pool = nullptr; }); });
        if (!g_pool) g_pool = new WorkPool();  // never destroyed: its threads end with the process
        p = g_pool;
    }
    p->run(n, f);
}

// Map fresh pages before the worker threads write into a buffer: dozens of threads faulting 4 KB pages
// of the same mapping in at once serialise on the process's memory-map lock (measured: 690 ms instead
// of 85 ms to inflate 200 MB on 8 threads).  The mappings ask for 2 MB pages, so a first pass touches
// one byte per 2 MB from a few threads (a few hundred faults, each zeroing 2 MB - bandwidth, not lock,
// bound); the 4 KB pass after it finds the pages present unless the kernel had no huge page to give.
inline void touch_pages(uint8_t* p, size_t from, size_t to) {
    constexpr size_t kStride = size_t(2) << 20;
    if (to > from && to - from >= 32 * kStride) {
        const size_t first = (from + kStride - 1) / kStride, last = (to + kStride - 1) / kStride;
        const int nt = 8;
        parallel_run(nt, [&](int t) {
            for (size_t c = first + (size_t)t; c < last; c += nt)
                if (c * kStride < to) p[c * kStride] = 0;
        });
    }
    for (size_t o = from; o < to; o += 4096) p[o] = 0;
}

// Large scratch buffers (the inflated text / BAM image) come from anonymous mappings that ask for
// transparent huge pages: a 650 MB text image is 160 K page faults on 4 KB pages (80+ ms on the one
// touching thread) and ~320 on 2 MB pages.
#if defined(__SANITIZE_ADDRESS__)
#define FTK_ASAN_BUILD 1
#elif defined(__has_feature)
#if __has_feature(address_sanitizer)
#define FTK_ASAN_BUILD 1
#endif
#endif
#if defined(FTK_ASAN_BUILD)
// sanitizer build: heap blocks of the exact size, so that the redzones stay around the parsers' inputs
inline size_t huge_round(size_t n) { return n; }
inline uint8_t* huge_map(size_t bytes) { return (uint8_t*)malloc(bytes ? bytes : 1); }
inline void huge_unmap(uint8_t* p, size_t) { free(p); }
inline uint8_t* huge_remap(uint8_t* p, size_t, size_t bytes) { return (uint8_t*)realloc(p, bytes); }
#else
constexpr size_t kHugePage = size_t(2) << 20;
inline size_t huge_round(size_t n) { return (n + kHugePage - 1) / kHugePage * kHugePage; }
inline uint8_t* huge_map(size_t bytes) {  // bytes: a multiple of kHugePage
    void* q = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (q == MAP_FAILED) return nullptr;
    (void)madvise(q, bytes, MADV_HUGEPAGE);
    return (uint8_t*)q;
}
inline void huge_unmap(uint8_t* p, size_t bytes) { munmap(p, bytes); }
inline uint8_t* huge_remap(uint8_t* p, size_t old_bytes, size_t bytes) {  // keeps the pages, no copy
    void* r = mremap(p, old_bytes, bytes, MREMAP_MAYMOVE);
    if (r == MAP_FAILED) return nullptr;
    (void)madvise(r, bytes, MADV_HUGEPAGE);
    return (uint8_t*)r;
}
#endif

struct Bytes {
    uint8_t* p = nullptr;
    size_t n = 0, mapped = 0;
    Bytes() = default;
    Bytes(const Bytes&) = delete;
    Bytes& operator=(const Bytes&) = delete;
    ~Bytes() { release(); }
    void release() {
        if (p) huge_unmap(p, mapped);
        p = nullptr;
        n = mapped = 0;
    }
    void alloc(size_t m) {
        release();
        mapped = huge_round(m ? m : 1);
        p = huge_map(mapped);
        if (!p) throw std::bad_alloc();
        n = m;
        touch_pages(p, 0, m);
    }
    uint8_t* data() { return p; }
    const uint8_t* data() const { return p; }
    size_t size() const { return n; }
};

struct Stopwatch {  // FTK_DECODE_TIMING=1 prints stage times to stderr
    bool on = getenv("FTK_DECODE_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[ftk decode] %-22s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
        t = now;
    }
};

struct Columns {
    std::vector<int32_t> start, end, r1s, r1e;
    std::vector<int32_t> ord;  // BAM: rank of the read1 record in the file (set by sort_by_start)
    std::vector<uint8_t> mapq, strand;
    // BAM: records the reference handles differently from a row (ftk_bamrule.h): [0] fragments the columns cannot
    // hold, [1] CIGAR-less read1 records with TLEN < 0
    uint32_t skipped[2] = {0, 0};
    void append(const Columns& o) {
        skipped[0] += o.skipped[0];
        skipped[1] += o.skipped[1];
        start.insert(start.end(), o.start.begin(), o.start.end());
        end.insert(end.end(), o.end.begin(), o.end.end());
        mapq.insert(mapq.end(), o.mapq.begin(), o.mapq.end());
        strand.insert(strand.end(), o.strand.begin(), o.strand.end());
        r1s.insert(r1s.end(), o.r1s.begin(), o.r1s.end());
        r1e.insert(r1e.end(), o.r1e.begin(), o.r1e.end());
    }
};

// Final, immutable form of one contig's columns: ONE block, page-locked when a
// HIP device is present (so the upload is a straight DMA), plain memory otherwise.
struct Packed {
    void* base = nullptr;
    bool pinned = false;
    size_t rows = 0;
    int32_t *start = nullptr, *end = nullptr, *r1s = nullptr, *r1e = nullptr, *ord = nullptr;
    uint8_t *mapq = nullptr, *strand = nullptr;
};

// Device blocks of the contigs parsed on the GPU are recycled: hipFree waits for ALL work on the device - the
// consumer's 2 GB copy-back included - so a producer that frees and allocates per contig would run in lock
// step with the consumer.  Any cached block that is large enough is reused (contigs shrink along a genome).
struct DeviceBlockCache {
    struct Blk { void* p; size_t cap; int device; };
    std::mutex mu;
    std::vector<Blk> free_list;
    size_t cached = 0;
    // smallest adequate block, or (largest = true: a contig's first block, which will grow) the largest
    void* take(size_t bytes, int device, size_t* cap_out, bool largest) {
        {
            std::lock_guard<std::mutex> lk(mu);
            int best = -1;
            for (int i = 0; i < (int)free_list.size(); ++i)
                if (free_list[i].device == device && free_list[i].cap >= bytes &&
                    (best < 0 || (largest ? free_list[i].cap > free_list[best].cap : free_list[i].cap < free_list[best].cap)))
                    best = i;
            if (best >= 0) {
                Blk b = free_list[best];
                free_list.erase(free_list.begin() + best);
                cached -= b.cap;
                *cap_out = b.cap;
                return b.p;
            }
        }
        void* p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        *cap_out = bytes;
        return p;
    }
    void give(void* p, size_t cap, int device) {
        if (!p) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (free_list.size() < 8 && cached + cap <= (size_t(4) << 30)) {
                free_list.push_back({p, cap, device});
                cached += cap;
                return;
            }
        }
        (void)hipFree(p);
    }
    size_t trim() {  // free every idle block; returns the bytes released
        std::vector<Blk> drop;
        {
            std::lock_guard<std::mutex> lk(mu);
            drop.swap(free_list);
            cached = 0;
        }
        size_t n = 0;
        for (auto& b : drop) {
            (void)hipSetDevice(b.device);
            (void)hipFree(b.p);
            n += b.cap;
        }
        return n;
    }
};
DeviceBlockCache& device_cache() {
    static DeviceBlockCache* c = new DeviceBlockCache();  // leaked: the driver frees at process exit
    return *c;
}
}  // namespace
namespace ftk_host {
void* device_block_take(size_t bytes, int device, size_t* cap_out) { return device_cache().take(bytes, device, cap_out, false); }
void device_block_give(void* p, size_t cap, int device) { device_cache().give(p, cap, device); }
}  // namespace ftk_host
namespace {

// A contig whose columns live in device memory (the streaming text decoder with the GPU row parser): ONE
// block (start | end | mapq | strand at the block's row capacity), grown piece by piece with device-to-device
// copies on the parse stream; `ready` is recorded behind the last copy and waited for by the consumer's stream.
struct DevColumns {
    void* base = nullptr;
    size_t bytes = 0;
    int32_t *start = nullptr, *end = nullptr;
    uint8_t *mapq = nullptr, *strand = nullptr;
    // BAM contigs (parsed on the device, ftk_bamparse.hip): the read1 span of every fragment and, once the rows
    // have been sorted by fragment start, their file-order rank
    bool bam = false;
    int32_t *r1s = nullptr, *r1e = nullptr, *ord = nullptr;
    size_t rows = 0, cap = 0;
    int device = 0;
    hipEvent_t ready = nullptr;
    DevColumns() = default;
    DevColumns(const DevColumns&) = delete;
    DevColumns& operator=(const DevColumns&) = delete;
    ~DevColumns() {
        (void)hipSetDevice(device);
        if (ready) {
            (void)hipEventSynchronize(ready);  // the block may be handed out again right away
            (void)hipEventDestroy(ready);
        } else if (base) {
            (void)hipDeviceSynchronize();  // a contig abandoned half way (error / close): copies may be in flight
        }
        device_cache().give(base, bytes, device);
    }
    size_t row_bytes() const { return bam ? 22 : 10; }
    // room for `more` rows; existing rows are moved on `s` (which is drained before the old block is given back)
    bool reserve(size_t more, hipStream_t s, bool exact = false) {
        if (rows + more <= cap) return true;
        // (a multiple of 64 rows: the arrays then start 256-byte aligned and the capacity computed back
        // from the block's size is never below the request)
        // A contig's first block takes four pieces' worth of rows (chr1 at 30x is four pieces): growing a block
        // means draining the parse stream - with the next piece's DMA and kernels already in it - and blocks are
        // recycled, so the generous first size is paid once.
        // (exact: a block that will not grow - the sorted copy of a finished BAM contig)
        const size_t first = rows == 0 && !exact ? 4 * more : 0;
        const size_t want = (std::max<size_t>(std::max(std::max(rows + more, exact ? 0 : 2 * cap), first), exact ? 64 : size_t(1) << 20) + 63) / 64 * 64;
        size_t got_bytes = 0;
        const size_t rb = row_bytes();
        void* nb = device_cache().take(want * rb + 1024, device, &got_bytes, rows == 0 && !exact);
        if (!nb) return false;
        const size_t ncap = (got_bytes - 1024) / rb / 64 * 64;  // rows the block holds
        if (ncap < rows + more) {  // cannot happen; never write past a block
            device_cache().give(nb, got_bytes, device);
            return false;
        }
        int32_t* ns = (int32_t*)nb;
        int32_t* ne = ns + ncap;
        int32_t *n1s = nullptr, *n1e = nullptr, *nord = nullptr;
        int32_t* tail = ne + ncap;
        if (bam) {
            n1s = tail;
            n1e = n1s + ncap;
            nord = n1e + ncap;
            tail = nord + ncap;
        }
        uint8_t* nq = (uint8_t*)tail;
        uint8_t* nt = nq + ncap;
        if (rows) {
            bool ok = hipMemcpyAsync(ns, start, rows * 4, hipMemcpyDeviceToDevice, s) == hipSuccess &&
                      hipMemcpyAsync(ne, end, rows * 4, hipMemcpyDeviceToDevice, s) == hipSuccess &&
                      hipMemcpyAsync(nq, mapq, rows, hipMemcpyDeviceToDevice, s) == hipSuccess &&
                      hipMemcpyAsync(nt, strand, rows, hipMemcpyDeviceToDevice, s) == hipSuccess;
            if (ok && bam)
                ok = hipMemcpyAsync(n1s, r1s, rows * 4, hipMemcpyDeviceToDevice, s) == hipSuccess &&
                     hipMemcpyAsync(n1e, r1e, rows * 4, hipMemcpyDeviceToDevice, s) == hipSuccess;
            ok = ok && hipStreamSynchronize(s) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                device_cache().give(nb, got_bytes, device);
                return false;
            }
        }
        device_cache().give(base, bytes, device);
        base = nb; bytes = got_bytes;
        start = ns; end = ne; mapq = nq; strand = nt;
        r1s = n1s; r1e = n1e; ord = nord;
        cap = ncap;
        return true;
    }
    // append n rows from device (kind D2D) or host (H2D; the call returns when the source may be released)
    bool append(const int32_t* s0, const int32_t* e0, const uint8_t* q0, const uint8_t* t0, size_t n, hipMemcpyKind kind,
                hipStream_t s, const int32_t* a0 = nullptr, const int32_t* b0 = nullptr) {
        if (!n) return true;
        if (!reserve(n, s)) return false;
        if (kind == hipMemcpyDeviceToDevice) {  // one launch instead of four to six copies (ftk_textparse.h)
            if (bam && !(a0 && b0)) return false;
            ftk::append_rows_launch(s, start + rows, end + rows, mapq + rows, strand + rows, bam ? r1s + rows : nullptr,
                                    bam ? r1e + rows : nullptr, s0, e0, q0, t0, a0, b0, n);
            if (hipGetLastError() != hipSuccess) return false;
            rows += n;
            return true;
        }
        bool ok = hipMemcpyAsync(start + rows, s0, n * 4, kind, s) == hipSuccess &&
                  hipMemcpyAsync(end + rows, e0, n * 4, kind, s) == hipSuccess &&
                  hipMemcpyAsync(mapq + rows, q0, n, kind, s) == hipSuccess &&
                  hipMemcpyAsync(strand + rows, t0, n, kind, s) == hipSuccess;
        if (ok && bam)
            ok = a0 && b0 && hipMemcpyAsync(r1s + rows, a0, n * 4, kind, s) == hipSuccess &&
                 hipMemcpyAsync(r1e + rows, b0, n * 4, kind, s) == hipSuccess;
        if (ok && kind == hipMemcpyHostToDevice) ok = hipStreamSynchronize(s) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); return false; }
        rows += n;
        return true;
    }
};

struct Contig {
    std::string name;
    int64_t length = -1;
    std::shared_ptr<DevColumns> dev;  // set instead of c / parts / p.base when the columns are device-resident
    Columns c;   // parse-time storage, emptied by pack()
    std::vector<Columns> parts;  // streaming text decoder: the contig's runs in order, packed without merging
    Packed p;
};

bool have_hip_device() {
    static const bool yes = [] {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return false; }
        return n > 0;
    }();
    return yes;
}

// Page-locked host memory, large blocks.  hipHostMalloc allocates AND pins page by page on one thread - 0.2 ms per MB,
// and concurrent calls serialise (8 x 96 MB: 174 ms one after the other, 171 ms from eight threads) - which was most of
// what a process's first stream paid (a 5.9 GB BAM: 0.53 s first, 0.28 s warm; 163 ms of it the eight slots' staging).
// The same memory from an anonymous mapping that asks for 2 MB pages, touched by a few threads, then REGISTERED
// (hipHostRegister only pins what is there) takes 7.6 ms for the 768 MB, copies at the same 57 GB/s in both directions,
// asynchronously like a hipHostMalloc block, and goes back in 11 ms instead of 45 (probe: docs/experiments.md,
// "Page-locking").  FTK_PINNED_VIA=malloc keeps hipHostMalloc; a block that cannot be registered falls back to it.
struct PinnedMaps {
    std::mutex mu;
    std::unordered_map<void*, size_t> mapped;  // registered mappings: base -> mapped bytes
};
PinnedMaps& pinned_maps() {
    static PinnedMaps* m = new PinnedMaps();
    return *m;
}
void* pinned_map(size_t bytes) {
    static const bool via_malloc = [] {
        const char* e = getenv("FTK_PINNED_VIA");
        return e && strcmp(e, "malloc") == 0;
    }();
    void* p = nullptr;
#if !defined(FTK_ASAN_BUILD)
    if (!via_malloc && bytes >= (size_t(1) << 20)) {
        const size_t m = huge_round(bytes);
        uint8_t* q = huge_map(m);
        if (q) {
            // every 4 KB page present before the driver walks them (a 2 MB page: one fault for 512 of them)
            const int nt = m >= (size_t(32) << 20) ? 8 : 1;
            auto touch = [q, m, nt](int t) {
                const size_t a = m / kHugePage * (size_t)t / (size_t)nt * kHugePage, b = m / kHugePage * (size_t)(t + 1) / (size_t)nt * kHugePage;
                for (size_t o = a; o < b; o += 4096) q[o] = 0;
            };
            std::vector<std::thread> th;
            for (int t = 1; t < nt; ++t) th.emplace_back(touch, t);
            touch(0);
            for (auto& x : th) x.join();
            if (hipHostRegister(q, m, hipHostRegisterDefault) == hipSuccess) {
                std::lock_guard<std::mutex> lk(pinned_maps().mu);
                pinned_maps().mapped[q] = m;
                return q;
            }
            (void)hipGetLastError();
            huge_unmap(q, m);
        }
    }
#endif
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return nullptr;
    }
    return p;
}
void pinned_unmap(void* p) {
    if (!p) return;
    size_t m = 0;
    {
        std::lock_guard<std::mutex> lk(pinned_maps().mu);
        auto it = pinned_maps().mapped.find(p);
        if (it != pinned_maps().mapped.end()) {
            m = it->second;
            pinned_maps().mapped.erase(it);
        }
    }
    if (!m) {
        (void)hipHostFree(p);
        return;
    }
    (void)hipHostUnregister(p);
    huge_unmap((uint8_t*)p, m);
}

// Page-locked blocks are recycled (a streamed file asks for one block per contig and frees it a moment later; a
// fresh block is cheap since pinned_map - ~0.01 ms per MB - but not free, and hipHostMalloc, its fall-back, pins at
// ~0.2 ms per MB and unpins about as slowly).  Two
// caches: the decoder's contig tables (a block is reused for a request of at least half its size, so that
// a small table does not sit on a huge block), and the callers' result arrays (ftk_host_alloc: any block
// that is large enough - results shrink from contig to contig and one block then serves them all).
struct PinnedCache {
    struct Blk { void* p; size_t cap; };
    const bool any_larger;   // reuse a block of any size >= the request
    const size_t max_bytes;  // held in the free list at most
    const size_t max_live;   // handed out at most (0: no limit); beyond it alloc() fails and the caller uses pageable memory
    const bool pinned;       // false: ordinary (pageable) memory, recycled the same way - see plain_result_cache()
    size_t live_bytes = 0;
    std::mutex mu;
    std::vector<Blk> free_list;
    std::vector<Blk> live;  // capacity of the blocks handed out (needed when they come back)
    size_t cached = 0;
    PinnedCache(bool any, size_t cap, size_t live_cap, bool pin = true)
        : any_larger(any), max_bytes(cap), max_live(live_cap), pinned(pin) {}

    void* fresh(size_t bytes) const {
        void* p = nullptr;
        if (pinned) return pinned_map(bytes);
        // 2 MB-aligned and advised huge: the first touch of a 2 GB result is then a thousand faults, not half a million
        if (posix_memalign(&p, size_t(2) << 20, bytes) != 0) return nullptr;
        (void)madvise(p, bytes, MADV_HUGEPAGE);
        return p;
    }
    void drop(void* p) const {
        if (pinned) pinned_unmap(p);
        else free(p);
    }

    void* alloc(size_t bytes) {
        {
            std::lock_guard<std::mutex> lk(mu);
            if (max_live && live_bytes + bytes > max_live) return nullptr;
            // smallest block that fits; a block may exceed the request by 2x (tables) or 8x + 64 MB (results:
            // they shrink from contig to contig, chr1's block serves chrY) -- never an 8 MB request on a 2 GB
            // block -- and what it pins in full must fit under the limit, since that is what gets accounted
            const size_t slack = any_larger ? 8 * bytes + (size_t(64) << 20) : 2 * bytes + (1 << 20);
            int best = -1;
            for (int i = 0; i < (int)free_list.size(); ++i)
                if (free_list[i].cap >= bytes && free_list[i].cap <= slack &&
                    (!max_live || live_bytes + free_list[i].cap <= max_live) &&
                    (best < 0 || free_list[i].cap < free_list[best].cap))
                    best = i;
            if (best >= 0) {
                Blk b = free_list[best];
                free_list.erase(free_list.begin() + best);
                cached -= b.cap;
                live.push_back(b);
                live_bytes += b.cap;
                return b.p;
            }
        }
        void* p = fresh(bytes);
        if (!p) return nullptr;
        std::lock_guard<std::mutex> lk(mu);
        live.push_back({p, bytes});
        live_bytes += bytes;
        return p;
    }

    bool release(void* p) {  // false: not one of this cache's blocks
        size_t cap = 0;
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < live.size(); ++i)
                if (live[i].p == p) { cap = live[i].cap; live.erase(live.begin() + i); break; }
            if (!cap) return false;
            live_bytes -= cap;
            if (free_list.size() < 8 && cached + cap <= max_bytes) {
                free_list.push_back({p, cap});
                cached += cap;
                return true;
            }
        }
        drop(p);
        return true;
    }
    size_t trim() {  // unpin every idle block; returns the bytes released
        std::vector<Blk> drop;
        {
            std::lock_guard<std::mutex> lk(mu);
            drop.swap(free_list);
            cached = 0;
        }
        size_t n = 0;
        for (auto& b : drop) {
            this->drop(b.p);
            n += b.cap;
        }
        return n;
    }
};
// leaked on purpose: the driver unpins at process exit
PinnedCache& table_cache() {
    static PinnedCache* c = new PinnedCache(false, size_t(2) << 30, 0);
    return *c;
}
PinnedCache& result_cache() {
    // at most 8 GB of result arrays are page-locked at a time (a caller that keeps every contig's per-base
    // scores would otherwise lock tens of GB); FTK_PINNED_RESULT_LIMIT_MB overrides
    static PinnedCache* c = [] {
        const char* e = getenv("FTK_PINNED_RESULT_LIMIT_MB");
        const long long mb = e ? atoll(e) : 8192;
        return new PinnedCache(true, size_t(6) << 30, (size_t)std::max<long long>(mb, 1) << 20);
    }();
    return *c;
}

// Result arrays the DEVICE never writes (ftk_host_alloc_pageable: per-base scores that cross the link as int16 and are
// widened into the array by the host threads): ordinary memory, kept between calls like the page-locked ones - a
// block that went back to the C library would be unmapped and faulted in again by every call.
PinnedCache& plain_result_cache() {
    static PinnedCache* c = new PinnedCache(true, size_t(6) << 30, 0, /*pin=*/false);
    return *c;
}

void* pinned_alloc(size_t bytes) { return table_cache().alloc(bytes); }

void pinned_free(void* p) {
    if (p) (void)table_cache().release(p);
}

// Lay a contig's final columns out inside an existing block (no copy).
size_t packed_bytes(size_t m, bool bam) {
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    return (bam ? 5 : 2) * up(m * 4 + 16) + 2 * up(m + 16);
}

void place(Packed& p, char* q, size_t m, bool bam) {
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t b32 = up(m * 4 + 16), b8 = up(m + 16);
    p.rows = m;
    p.start = (int32_t*)q; q += b32;
    p.end = (int32_t*)q; q += b32;
    if (bam) { p.r1s = (int32_t*)q; q += b32; p.r1e = (int32_t*)q; q += b32; p.ord = (int32_t*)q; q += b32; }
    p.mapq = (uint8_t*)q; q += b8;
    p.strand = (uint8_t*)q;
}

void pack(Contig& ct) {
    Columns& c = ct.c;
    const size_t m = c.start.size();
    const bool bam = !c.r1s.empty();
    auto up = [](size_t v) { return (v + 255) / 256 * 256; };
    const size_t b32 = up(m * 4 + 16), b8 = up(m + 16);
    const size_t total = (bam ? 5 : 2) * b32 + 2 * b8;
    Packed& p = ct.p;
    p.rows = m;
    if (have_hip_device() && (p.base = pinned_alloc(total)) != nullptr) {
        p.pinned = true;
    } else {
        p.base = malloc(total);
        p.pinned = false;
    }
    if (!p.base) return;
    char* q = (char*)p.base;
    p.start = (int32_t*)q; q += b32;
    p.end = (int32_t*)q; q += b32;
    if (bam) { p.r1s = (int32_t*)q; q += b32; p.r1e = (int32_t*)q; q += b32; p.ord = (int32_t*)q; q += b32; }
    p.mapq = (uint8_t*)q; q += b8;
    p.strand = (uint8_t*)q;
    if (m) {
        memcpy(p.start, c.start.data(), m * 4);
        memcpy(p.end, c.end.data(), m * 4);
        memcpy(p.mapq, c.mapq.data(), m);
        memcpy(p.strand, c.strand.data(), m);
        if (bam) {
            memcpy(p.r1s, c.r1s.data(), m * 4);
            memcpy(p.r1e, c.r1e.data(), m * 4);
            if (c.ord.size() == m) memcpy(p.ord, c.ord.data(), m * 4);
            else for (size_t i = 0; i < m; ++i) p.ord[i] = (int32_t)i;
        }
    }
    c = Columns{};
}

}  // namespace

namespace ftk_host {

void parallel_run(int n, const std::function<void(int)>& fn) { ::parallel_run(n, fn); }

void parallel_run_results(int n, const std::function<void(int)>& fn) {
    static std::mutex init;
    static WorkPool* pool = nullptr;
    static std::once_flag fork_once;
    WorkPool* p;
    {
        std::lock_guard<std::mutex> lk(init);
        std::call_once(fork_once, [] { pthread_atfork(nullptr, nullptr, [] { pool = nullptr; }); });
        if (!pool) pool = new WorkPool();  // never destroyed: its threads end with the process
        p = pool;
    }
    p->run(n, fn);
}

void set_decode_error(const char* msg) { g_decode_err = msg ? msg : ""; }

int default_threads() {
    static const int n = [] {
        int c = (int)std::thread::hardware_concurrency();
        cpu_set_t set;
        if (sched_getaffinity(0, sizeof(set), &set) == 0) c = CPU_COUNT(&set);
        if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2 quota: "max" or "<quota> <period>"
            char q[32];
            long long period = 0;
            if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0)
                c = std::min<long long>(c, std::max<long long>(1, (atoll(q) + period / 2) / period));
            fclose(f);
        }
        // one rank per GPU on one node: every rank takes its share of the host cores (LOCAL_WORLD_SIZE is set by
        // torchrun and by sharding.launch_ranks), so 8 ranks do not start 8 x 16 threads on a 16-core quota;
        // FTK_HOST_THREADS overrides
        if (const char* w = getenv("LOCAL_WORLD_SIZE"))
            if (atoi(w) > 1) c = std::max(1, c / atoi(w));
        if (const char* t = getenv("FTK_HOST_THREADS"))
            if (atoi(t) > 0) c = atoi(t);
        return std::max(1, std::min(c, 64));
    }();
    return n;
}

}  // namespace ftk_host

struct ftk_fragtable {
    std::vector<Contig> contigs;
    bool bed6 = false;
    bool bam = false;
    int64_t skipped[2] = {0, 0};  // ftk_fragtable_skipped (whole-file BAM decoder)
    void* block = nullptr;  // one allocation holding every contig's columns (text decoder)
    bool block_pinned = false;
    ~ftk_fragtable() {
        if (block) { if (block_pinned) pinned_free(block); else free(block); }
        for (auto& ct : contigs) {
            if (!ct.p.base) continue;
            if (ct.p.pinned) pinned_free(ct.p.base); else free(ct.p.base);
        }
    }
};

namespace {

// Whole file -> memory.  Large files are read as a few concurrent pread streams: one thread copies out
// of the page cache at ~8 GB/s, and a cold file gets several requests in flight.
bool read_file(const char* path, Bytes* out) {
    const int fd = open(path, O_RDONLY | O_CLOEXEC);
    if (fd < 0) return false;
    struct stat st;
    if (fstat(fd, &st) != 0 || !S_ISREG(st.st_mode) || st.st_size < 0) { close(fd); return false; }
    const size_t sz = (size_t)st.st_size;
    out->alloc(sz);
    auto read_span = [&](size_t from, size_t to) {
        while (from < to) {
            const ssize_t got = pread(fd, out->data() + from, to - from, (off_t)from);
            if (got <= 0) return false;
            from += (size_t)got;
        }
        return true;
    };
    bool ok = true;
    constexpr size_t kSpan = size_t(8) << 20;
    if (sz < 4 * kSpan) {
        ok = read_span(0, sz);
    } else {
        const int nt = 4;
        std::atomic<size_t> next{0};
        std::atomic<int> bad{0};
        parallel_run(nt, [&](int) {
            for (;;) {
                const size_t from = next.fetch_add(kSpan);
                if (from >= sz || bad.load()) break;
                if (!read_span(from, std::min(sz, from + kSpan))) bad = 1;
            }
        });
        ok = !bad.load();
    }
    close(fd);
    return ok;
}

struct Block {
    size_t in_off, in_len;   // raw deflate payload
    size_t out_off, out_len;
};

// Parse one gzip member header; returns payload offset or 0 on error.
// *bsize = BGZF total block size when the BC subfield is present, else 0.
size_t gzip_header(const uint8_t* p, size_t n, size_t off, size_t* bsize) {
    *bsize = 0;
    if (off + 18 > n) return 0;
    if (p[off] != 31 || p[off + 1] != 139 || p[off + 2] != 8) return 0;
    int flg = p[off + 3];
    size_t q = off + 10;
    if (flg & 4) {
        if (q + 2 > n) return 0;
        size_t xlen = p[q] | (p[q + 1] << 8);
        q += 2;
        if (q + xlen > n) return 0;
        size_t x = q, xe = q + xlen;
        while (x + 4 <= xe) {
            size_t slen = p[x + 2] | (p[x + 3] << 8);
            if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2 && x + 6 <= xe) *bsize = (size_t)(p[x + 4] | (p[x + 5] << 8)) + 1;
            x += 4 + slen;
        }
        q = xe;
    }
    if (flg & 8) { while (q < n && p[q]) ++q; ++q; }
    if (flg & 16) { while (q < n && p[q]) ++q; ++q; }
    if (flg & 2) q += 2;
    return q <= n ? q : 0;
}

// Raw-deflate payload of one BGZF block -> exactly out_len bytes.  libdeflate (present in the image as
// a runtime library without headers, so bound with dlopen; its three entry points have been stable
// since 1.0) inflates 2-3x faster than zlib; zlib with one reused stream per worker is the other path.
// FTK_NO_LIBDEFLATE=1 forces zlib.
struct DeflateLib {
    void* (*alloc)() = nullptr;
    int (*run)(void*, const void*, size_t, void*, size_t, size_t*) = nullptr;
    void (*release)(void*) = nullptr;
    uint32_t (*crc)(uint32_t, const void*, size_t) = nullptr;  // libdeflate_crc32 (several GB/s per thread), else zlib's
    DeflateLib() {
        if (getenv("FTK_NO_LIBDEFLATE")) return;
        void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
        if (!h) return;
        auto a = (void* (*)())dlsym(h, "libdeflate_alloc_decompressor");
        auto r = (int (*)(void*, const void*, size_t, void*, size_t, size_t*))dlsym(h, "libdeflate_deflate_decompress");
        auto f = (void (*)(void*))dlsym(h, "libdeflate_free_decompressor");
        if (a && r && f) { alloc = a; run = r; release = f; }
        crc = (uint32_t (*)(uint32_t, const void*, size_t))dlsym(h, "libdeflate_crc32");
    }
};

const DeflateLib& deflate_lib() {
    static const DeflateLib lib;
    return lib;
}
inline uint32_t crc32_of(const uint8_t* p, size_t n) {
    if (deflate_lib().crc) return deflate_lib().crc(0, p, n);
    return (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n);
}
inline uint32_t trailer_crc(const uint8_t* t) { return (uint32_t)t[0] | ((uint32_t)t[1] << 8) | ((uint32_t)t[2] << 16) | ((uint32_t)t[3] << 24); }

class BlockInflater {
    void* fast = nullptr;
    z_stream zs;
    bool z_ready = false;

public:
    BlockInflater() {
        memset(&zs, 0, sizeof(zs));
        if (deflate_lib().alloc) fast = deflate_lib().alloc();
    }
    BlockInflater(const BlockInflater&) = delete;
    BlockInflater& operator=(const BlockInflater&) = delete;
    ~BlockInflater() {
        if (fast) deflate_lib().release(fast);
        if (z_ready) inflateEnd(&zs);
    }
    bool operator()(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len) {
        if (fast) {
            size_t got = 0;
            return deflate_lib().run(fast, in, in_len, out, out_len, &got) == 0 && got == out_len;
        }
        if (!z_ready) {
            if (inflateInit2(&zs, -15) != Z_OK) return false;
            z_ready = true;
        } else if (inflateReset(&zs) != Z_OK) {
            return false;
        }
        zs.next_in = const_cast<Bytef*>(in);
        zs.avail_in = (uInt)in_len;
        zs.next_out = out;
        zs.avail_out = (uInt)out_len;
        return inflate(&zs, Z_FINISH) == Z_STREAM_END && zs.avail_out == 0;
    }
};

// own_threads: threads of this call's own (the pool runs one region at a time; a job beside the producer's regions
// must not hold it for the length of a whole piece)
// check_crc: compare every block's data with the CRC-32 of its gzip trailer (the 4 bytes behind the payload)
int inflate_block_list(const uint8_t* p, const std::vector<Block>& blocks, int n_threads, uint8_t* out, bool own_threads = false,
                       bool check_crc = false) {
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    auto work = [&]() {
        BlockInflater inf;
        for (;;) {
            // a few blocks per grab: 64 KB blocks finish in ~30 us and the counter is shared by all workers
            const size_t i0 = next.fetch_add(4);
            if (i0 >= blocks.size() || bad.load()) break;
            for (size_t i = i0; i < std::min(i0 + 4, blocks.size()); ++i) {
                const Block& b = blocks[i];
                if (b.out_len == 0) continue;
                if (!inf(p + b.in_off, b.in_len, out + b.out_off, b.out_len)) { bad = 1; break; }
                if (check_crc && crc32_of(out + b.out_off, b.out_len) != trailer_crc(p + b.in_off + b.in_len)) { bad = 1; break; }
            }
        }
    };
    int nt = std::max(1, std::min<int>(n_threads, (int)blocks.size()));
    if (own_threads) {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work);
        work();
        for (auto& t : th) t.join();
    } else {
        parallel_run(nt, [&](int) { work(); });
    }
    return bad.load() ? FTK_ERR_FORMAT : FTK_OK;
}

// Inflate a BGZF (block-parallel) or plain gzip (serial) file image.
int inflate_all(const Bytes& in, int n_threads, Bytes* out) {
    const uint8_t* p = in.data();
    const size_t n = in.size();
    if (n == 0) { out->alloc(0); return FTK_OK; }
    size_t bsize = 0;
    size_t pay = gzip_header(p, n, 0, &bsize);
    if (!pay) return dfail(FTK_ERR_FORMAT, "not a gzip/BGZF file");
    if (bsize) {
        std::vector<Block> blocks;
        size_t off = 0, total = 0;
        while (off < n) {
            size_t bs = 0;
            size_t q = gzip_header(p, n, off, &bs);
            if (!q || !bs || off + bs > n || q + 8 > off + bs) return dfail(FTK_ERR_FORMAT, "corrupt BGZF block at %zu", off);
            const uint8_t* tr = p + off + bs - 8;
            size_t isize = (size_t)tr[4] | ((size_t)tr[5] << 8) | ((size_t)tr[6] << 16) | ((size_t)tr[7] << 24);
            blocks.push_back({q, off + bs - 8 - q, total, isize});
            total += isize;
            off += bs;
        }
        out->alloc(total);
        if (inflate_block_list(p, blocks, n_threads, out->data()) != FTK_OK) return dfail(FTK_ERR_FORMAT, "BGZF inflate failed");
        return FTK_OK;
    }
    // plain (possibly multi-member) gzip
    std::vector<uint8_t> acc;
    z_stream zs;
    memset(&zs, 0, sizeof(zs));
    if (inflateInit2(&zs, 15 + 16) != Z_OK) return dfail(FTK_ERR_FORMAT, "zlib init failed");
    zs.next_in = const_cast<Bytef*>(p);
    zs.avail_in = (uInt)std::min<size_t>(n, 0xFFFFFFFFu);
    std::vector<uint8_t> buf(1 << 20);
    for (;;) {
        zs.next_out = buf.data();
        zs.avail_out = (uInt)buf.size();
        int rc = inflate(&zs, Z_NO_FLUSH);
        if (rc != Z_OK && rc != Z_STREAM_END) { inflateEnd(&zs); return dfail(FTK_ERR_FORMAT, "gzip inflate failed (%d)", rc); }
        acc.insert(acc.end(), buf.data(), buf.data() + (buf.size() - zs.avail_out));
        if (rc == Z_STREAM_END) {
            if (zs.avail_in == 0) break;
            if (inflateReset(&zs) != Z_OK) { inflateEnd(&zs); return dfail(FTK_ERR_FORMAT, "gzip member reset failed"); }
        } else if (zs.avail_in == 0 && zs.avail_out != 0) {
            break;  // truncated stream: keep what we have
        }
    }
    inflateEnd(&zs);
    out->alloc(acc.size());
    if (!acc.empty()) memcpy(out->data(), acc.data(), acc.size());
    return FTK_OK;
}

// Python int(): optional surrounding blanks, optional sign, decimal digits.
bool parse_int(const char* b, const char* e, long long* v) {
    while (b < e && (*b == ' ' || *b == '\r')) ++b;
    while (e > b && (e[-1] == ' ' || e[-1] == '\r')) --e;
    if (b == e) return false;
    bool neg = false;
    if (*b == '+' || *b == '-') { neg = (*b == '-'); ++b; }
    if (b == e) return false;
    long long x = 0;
    for (; b < e; ++b) {
        if (*b < '0' || *b > '9') return false;
        x = x * 10 + (*b - '0');
        if (x > (1LL << 40)) return false;
    }
    *v = neg ? -x : x;
    return true;
}

struct Run {
    std::string name;
    Columns c;
};

// 1..10 decimal digits followed by `term`; returns the byte after `term`, nullptr for anything else.
// Stops at the first non-digit, so the caller only has to guarantee a '\n' somewhere ahead.
inline const char* plain_digits(const char* p, char term, uint64_t* v) {
    const char* s = p;
    uint64_t x = 0;
    unsigned d;
    while ((d = (unsigned)(unsigned char)*p - (unsigned)'0') <= 9) { x = x * 10 + d; ++p; }
    if (p == s || p - s > 10 || *p != term) return nullptr;
    *v = x;
    return p + 1;
}

// The row nearly every line is: the current contig's name, plain unsigned decimal fields, a one-character
// strand and the line end.  One forward pass, no per-field searches.  Returns the start of the next line,
// or nullptr when the general parser below has to look at the row (other contig, signs, blanks, a longer
// strand field, more columns, 11+ digits ...).  `lim` points at a '\n' at or after `p`.
inline const char* plain_row(const char* p, const char* lim, const std::string& name, bool bed6, Columns& c) {
    const size_t cl = name.size();
    if ((size_t)(lim - p) <= cl || memcmp(p, name.data(), cl) != 0 || p[cl] != '\t') return nullptr;
    p += cl + 1;
    uint64_t s, t, m;
    if (!(p = plain_digits(p, '\t', &s)) || !(p = plain_digits(p, '\t', &t))) return nullptr;
    if (bed6) {  // column 3 (a name) is not read
        while (*p != '\t' && *p != '\n') ++p;
        if (*p != '\t') return nullptr;
        ++p;
    }
    if (!(p = plain_digits(p, '\t', &m))) return nullptr;
    const char strand = *p;
    if (strand == '\n' || strand == '\t' || strand == '\r') return nullptr;
    ++p;
    if (*p == '\r') ++p;
    if (*p != '\n') return nullptr;
    if (s <= (uint64_t)INT32_MAX && t <= (uint64_t)INT32_MAX) {
        c.start.push_back((int32_t)s);
        c.end.push_back((int32_t)t);
        c.mapq.push_back((uint8_t)std::min<uint64_t>(m, 255));
        c.strand.push_back(strand == '+' ? 1 : 0);
    }
    return p + 1;
}

void parse_segment(const char* b, const char* e, bool bed6, const char* only, std::vector<Run>* runs) {
    const int mq_col = bed6 ? 4 : 3, st_col = bed6 ? 5 : 4;
    const size_t only_len = only ? strlen(only) : 0;
    Run* cur = nullptr;
    // last line end of the segment: rows before it may be scanned without a bound per byte
    const char* lim = e;
    while (lim > b && lim[-1] != '\n') --lim;
    lim = lim > b ? lim - 1 : nullptr;
    while (b < e) {
        if (cur && lim && b <= lim) {
            const char* next = plain_row(b, lim, cur->name, bed6, cur->c);
            if (next) { b = next; continue; }
        }
        const char* nl = (const char*)memchr(b, '\n', (size_t)(e - b));
        const char* le = nl ? nl : e;
        const char* line = b;
        b = nl ? nl + 1 : e;
        if (le > line && le[-1] == '\r') --le;
        if (le == line || *line == '#') continue;
        const char* fb[7];
        const char* fe[7];
        int nf = 0;
        const char* q = line;
        while (nf < 7) {
            const char* tab = (const char*)memchr(q, '\t', (size_t)(le - q));
            fb[nf] = q;
            fe[nf] = tab ? tab : le;
            ++nf;
            if (!tab) break;
            q = tab + 1;
        }
        if (nf <= st_col) continue;  // IndexError in the reference -> row skipped
        size_t cl = (size_t)(fe[0] - fb[0]);
        if (only && (cl != only_len || memcmp(fb[0], only, cl) != 0)) continue;
        long long s, t, m;
        if (!parse_int(fb[1], fe[1], &s) || !parse_int(fb[2], fe[2], &t) || !parse_int(fb[mq_col], fe[mq_col], &m)) continue;
        if (s < 0 || t < 0 || s > INT32_MAX || t > INT32_MAX || m < 0) continue;
        if (!cur || cur->name.size() != cl || memcmp(cur->name.data(), fb[0], cl) != 0) {
            runs->push_back(Run{std::string(fb[0], cl), {}});
            cur = &runs->back();
            // room for the rest of the segment at ~20 bytes a row: no regrowth copies in the common case
            const size_t guess = (size_t)(e - line) / 20 + 16;
            cur->c.start.reserve(guess);
            cur->c.end.reserve(guess);
            cur->c.mapq.reserve(guess);
            cur->c.strand.reserve(guess);
        }
        cur->c.start.push_back((int32_t)s);
        cur->c.end.push_back((int32_t)t);
        cur->c.mapq.push_back((uint8_t)std::min<long long>(m, 255));
        cur->c.strand.push_back(memchr(fb[st_col], '+', (size_t)(fe[st_col] - fb[st_col])) ? 1 : 0);
    }
}

Contig* find_or_add(ftk_fragtable* t, const std::string& name) {
    for (auto& c : t->contigs)
        if (c.name == name) return &c;
    t->contigs.push_back(Contig{name, -1, {}});
    return &t->contigs.back();
}

// Sort 64-bit keys: one chunk per thread, then pairwise merges level by level.  Returns the thread count used.
// Keys are (fragment start << 32 | file rank).  Large inputs go through one counting pass on 4 096-base buckets of
// the start coordinate (per-thread histograms, a stable scatter) and a small sort per bucket, every step on all
// threads: 2 passes over the keys instead of a chunk sort and log2(threads) merge levels of which the last run on one
// or two threads (4.8 M keys of a 60x BAM slice on 16 threads: 25 -> 6 ms).
bool bucket_sort_keys(uint64_t* key, size_t m, int nt) {
    Stopwatch sw;
    constexpr int kShift = 32 + 12;
    uint64_t top = 0;
    {
        std::vector<uint64_t> tmax(nt, 0);
        parallel_run(nt, [&](int t) {
            uint64_t mx = 0;
            for (size_t i = m * (size_t)t / (size_t)nt, e = m * (size_t)(t + 1) / (size_t)nt; i < e; ++i) mx = std::max(mx, key[i]);
            tmax[t] = mx;
        });
        for (uint64_t v : tmax) top = std::max(top, v);
    }
    sw.lap("    sort: max");
    const size_t nb = (size_t)(top >> kShift) + 1;
    if (nb > (size_t(1) << 20) || nb * (size_t)nt > m) return false;  // few keys per bucket: the histograms would dominate
    std::vector<uint32_t> cnt(nb * (size_t)nt, 0);
    parallel_run(nt, [&](int t) {
        uint32_t* c = cnt.data() + nb * (size_t)t;
        for (size_t i = m * (size_t)t / (size_t)nt, e = m * (size_t)(t + 1) / (size_t)nt; i < e; ++i) ++c[key[i] >> kShift];
    });
    sw.lap("    sort: count");
    std::vector<size_t> first(nb + 1);
    size_t run = 0;
    for (size_t b = 0; b < nb; ++b) {  // bucket by bucket, thread by thread inside: stable
        first[b] = run;
        for (int t = 0; t < nt; ++t) {
            const uint32_t c = cnt[nb * (size_t)t + b];
            cnt[nb * (size_t)t + b] = (uint32_t)(run - first[b]);  // the thread's offset inside the bucket
            run += c;
        }
    }
    first[nb] = run;
    sw.lap("    sort: offsets");
    std::unique_ptr<uint64_t[]> tmp(new uint64_t[m]);  // (not zeroed: the scatter's threads touch its pages first)
    parallel_run(nt, [&](int t) {
        uint32_t* c = cnt.data() + nb * (size_t)t;
        for (size_t i = m * (size_t)t / (size_t)nt, e = m * (size_t)(t + 1) / (size_t)nt; i < e; ++i) {
            const size_t b = (size_t)(key[i] >> kShift);
            tmp[first[b] + c[b]++] = key[i];
        }
    });
    sw.lap("    sort: scatter");
    parallel_run(nt, [&](int t) {  // buckets dealt by position in the output: equal shares of the keys
        const size_t lo = m * (size_t)t / (size_t)nt, hi = m * (size_t)(t + 1) / (size_t)nt;
        size_t b = (size_t)(std::upper_bound(first.begin(), first.end(), lo) - first.begin());
        if (b) --b;
        if (first[b] < lo) ++b;  // a bucket belongs to the thread its first key falls to
        for (; b < nb && first[b] < hi; ++b) {
            uint64_t* a = tmp.get() + first[b];
            uint64_t* z = tmp.get() + first[b + 1];
            if (z - a > 1 && !std::is_sorted(a, z)) std::sort(a, z);
            std::copy(a, z, key + first[b]);
        }
    });
    sw.lap("    sort: buckets");
    return true;
}

int sort_keys(uint64_t* key, size_t m, int n_threads) {
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(n_threads, 1), m / 65536));
    if (nt == 1) {
        std::sort(key, key + m);
        return nt;
    }
    if (bucket_sort_keys(key, m, nt)) return nt;
    std::unique_ptr<uint64_t[]> tmp(new uint64_t[m]);
    std::vector<size_t> cut(nt + 1);
    for (int t = 0; t <= nt; ++t) cut[t] = m * (size_t)t / (size_t)nt;
    parallel_run(nt, [&](int t) { std::sort(key + cut[t], key + cut[t + 1]); });
    uint64_t* src = key;
    uint64_t* dst = tmp.get();
    while (cut.size() > 2) {  // merge neighbours; an odd last chunk is copied through
        const size_t n_chunks = cut.size() - 1, n_pairs = n_chunks / 2;
        parallel_run((int)((n_chunks + 1) / 2), [&](int t) {
            const size_t a = cut[2 * t], mid = cut[2 * t + 1];
            if ((size_t)t < n_pairs) std::merge(src + a, src + mid, src + mid, src + cut[2 * t + 2], dst + a);
            else std::copy(src + a, src + mid, dst + a);
        });
        std::vector<size_t> next;
        for (size_t k = 0; k < cut.size(); k += 2) next.push_back(cut[k]);
        if (next.back() != m) next.push_back(m);
        cut.swap(next);
        std::swap(src, dst);
    }
    if (src != key) memcpy(key, src, m * sizeof(uint64_t));
    return nt;
}

// Stable order by fragment start.  BAM fragments arrive in read1-position order, i.e. nearly sorted
// (a reverse-strand read1 sits at the far end of its fragment), millions per contig: the keys
// (start << 32 | file rank) are sorted in one chunk per thread, the chunks merged pairwise level by
// level, and the columns gathered in parallel.
void sort_by_start(Columns& c, int n_threads = 1) {
    const size_t m = c.start.size();
    const bool bam_cols = !c.r1s.empty() || m == 0;
    if (bam_cols && c.ord.size() != m) {  // file order of the read1 records, to restore pysam's iteration order
        c.ord.resize(m);
        std::iota(c.ord.begin(), c.ord.end(), 0);
    }
    if (std::is_sorted(c.start.begin(), c.start.end())) return;
    std::vector<uint64_t> key(m);
    for (size_t i = 0; i < m; ++i) key[i] = ((uint64_t)(uint32_t)c.start[i] << 32) | (uint64_t)i;  // starts are >= 0
    const int nt = sort_keys(key.data(), m, n_threads);
    Columns s;
    const bool r1 = !c.r1s.empty();
    s.start.resize(m); s.end.resize(m); s.mapq.resize(m); s.strand.resize(m);
    if (r1) { s.r1s.resize(m); s.r1e.resize(m); s.ord.resize(m); }
    parallel_run(nt, [&](int t) {
        for (size_t i = m * (size_t)t / (size_t)nt, e = m * (size_t)(t + 1) / (size_t)nt; i < e; ++i) {
            const uint32_t j = (uint32_t)key[i];
            s.start[i] = c.start[j]; s.end[i] = c.end[j]; s.mapq[i] = c.mapq[j]; s.strand[i] = c.strand[j];
            if (r1) { s.r1s[i] = c.r1s[j]; s.r1e[i] = c.r1e[j]; s.ord[i] = c.ord[j]; }
        }
    });
    c = std::move(s);
}

inline int32_t rd_i32(const uint8_t* p) { return (int32_t)((uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24)); }
inline uint32_t rd_u32(const uint8_t* p) { return (uint32_t)rd_i32(p); }
inline uint16_t rd_u16(const uint8_t* p) { return (uint16_t)(p[0] | (p[1] << 8)); }

// One BAM alignment record -> fragment columns (io/alignment.py:60-71,242-268: the rule lives in ftk_bamrule.h,
// shared with the device parser); false = no row.  Records the reference treats differently from "no fragment"
// are counted in c.skipped.
inline bool bam_record(const uint8_t* r, uint32_t bs, Columns& c) {
    ftk::BamRow f;
    const int rule = ftk::bam_rule(r, bs, [](const uint8_t* p) { return rd_u32(p); }, f);
    if (rule != ftk::kBamFragment) {
        if (rule != ftk::kBamNotFragment) ++c.skipped[rule == ftk::kBamNoCigarReverse];
        return false;
    }
    c.start.push_back(f.fs);
    c.end.push_back(f.fe);
    c.mapq.push_back(f.mapq);
    c.strand.push_back(f.fwd);
    c.r1s.push_back(f.r1s);
    c.r1e.push_back(f.r1e);
    return true;
}

}  // namespace

extern "C" {

const char* ftk_fragtable_error(void) { return g_decode_err.c_str(); }

static int ftk_fragfile_decode_impl(const char* path, const char* contig, int n_threads, ftk_fragtable** out) {
    if (!path || !out) return dfail(FTK_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_threads < 1) n_threads = 1;
    Stopwatch sw;
    Bytes raw, text;
    if (!read_file(path, &raw)) return dfail(FTK_ERR_IO, "cannot read %s", path);
    sw.lap("read file");
    int rc = inflate_all(raw, n_threads, &text);
    if (rc) return rc;
    raw.alloc(0);
    sw.lap("inflate");
    std::unique_ptr<ftk_fragtable> t(new ftk_fragtable());
    const char* b = (const char*)text.data();
    const char* e = b + text.size();
    // layout detection on the first data row (io/alignment.py:143-156)
    {
        const char* q = b;
        while (q < e) {
            const char* nl = (const char*)memchr(q, '\n', (size_t)(e - q));
            const char* le = nl ? nl : e;
            if (le > q && *q != '#') {
                int tabs = 0;
                for (const char* x = q; x < le; ++x) tabs += (*x == '\t');
                t->bed6 = (tabs + 1) > 5;
                break;
            }
            if (!nl) break;
            q = nl + 1;
        }
    }
    // split at line boundaries, parse segments in parallel, merge in order
    int nseg = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, text.size() / (1 << 16) + 1));
    std::vector<const char*> cut(nseg + 1);
    cut[0] = b;
    cut[nseg] = e;
    for (int i = 1; i < nseg; ++i) {
        const char* q = b + text.size() * (size_t)i / (size_t)nseg;
        if (q < cut[i - 1]) q = cut[i - 1];
        const char* nl = (const char*)memchr(q, '\n', (size_t)(e - q));
        cut[i] = nl ? nl + 1 : e;
    }
    std::vector<std::vector<Run>> seg_runs(nseg);
    parallel_run(nseg, [&](int i) { parse_segment(cut[i], cut[i + 1], t->bed6, contig, &seg_runs[i]); });
    sw.lap("parse");
    // Assemble: row offsets of every run (serial, tiny), ONE block for the whole table (page-locked
    // when a HIP device is present), then every segment copies its runs to their final place in parallel.
    struct Dest { int contig; size_t off; };
    std::vector<std::vector<Dest>> dest(nseg);
    std::vector<size_t> rows_of;
    for (int sg = 0; sg < nseg; ++sg)
        for (auto& r : seg_runs[sg]) {
            Contig* ct = find_or_add(t.get(), r.name);
            const int ci = (int)(ct - t->contigs.data());
            if ((size_t)ci >= rows_of.size()) rows_of.resize(ci + 1, 0);
            dest[sg].push_back({ci, rows_of[ci]});
            rows_of[ci] += r.c.start.size();
        }
    size_t total = 0;
    std::vector<size_t> base_of(t->contigs.size());
    for (size_t ci = 0; ci < t->contigs.size(); ++ci) { base_of[ci] = total; total += packed_bytes(rows_of[ci], false); }
    if (total == 0) total = 256;
    if (have_hip_device() && (t->block = pinned_alloc(total)) != nullptr) {
        t->block_pinned = true;
    } else {
        t->block = malloc(total);
    }
    if (!t->block) return dfail(FTK_ERR_OOM, "out of host memory (%zu bytes)", total);
    for (size_t ci = 0; ci < t->contigs.size(); ++ci) {
        place(t->contigs[ci].p, (char*)t->block + base_of[ci], rows_of[ci], false);
        t->contigs[ci].p.pinned = t->block_pinned;
    }
    sw.lap("allocate (pinned)");
    auto assemble = [&](int sg) {
        for (size_t k = 0; k < seg_runs[sg].size(); ++k) {
            const Columns& c = seg_runs[sg][k].c;
            const Packed& p = t->contigs[dest[sg][k].contig].p;
            const size_t o = dest[sg][k].off, m = c.start.size();
            if (!m) continue;
            memcpy(p.start + o, c.start.data(), m * 4);
            memcpy(p.end + o, c.end.data(), m * 4);
            memcpy(p.mapq + o, c.mapq.data(), m);
            memcpy(p.strand + o, c.strand.data(), m);
        }
        std::vector<Run>().swap(seg_runs[sg]);
    };
    parallel_run(nseg, [&](int i) { assemble(i); });
    sw.lap("assemble");
    *out = t.release();
    return FTK_OK;
}

static int ftk_bam_decode_impl(const char* path, const char* contig, int n_threads, ftk_fragtable** out) {
    if (!path || !out) return dfail(FTK_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (n_threads < 1) n_threads = 1;
    Bytes raw, bam;
    Stopwatch sw;
    if (!read_file(path, &raw)) return dfail(FTK_ERR_IO, "cannot read %s", path);
    sw.lap("read file");
    int rc = inflate_all(raw, n_threads, &bam);
    if (rc) return rc;
    raw.alloc(0);
    sw.lap("inflate");
    const uint8_t* p = bam.data();
    const size_t n = bam.size();
    if (n < 12 || memcmp(p, "BAM\1", 4) != 0) return dfail(FTK_ERR_FORMAT, "%s is not a BAM file", path);
    size_t off = 4;
    uint32_t l_text = rd_u32(p + off);
    off += 4 + (size_t)l_text;
    if (off + 4 > n) return dfail(FTK_ERR_FORMAT, "truncated BAM header");
    uint32_t n_ref = rd_u32(p + off);
    off += 4;
    std::unique_ptr<ftk_fragtable> t(new ftk_fragtable());
    t->bam = true;
    std::vector<int> ref_to_contig(n_ref, -1);
    for (uint32_t r = 0; r < n_ref; ++r) {
        if (off + 4 > n) return dfail(FTK_ERR_FORMAT, "truncated BAM reference list");
        uint32_t l_name = rd_u32(p + off);
        off += 4;
        if (off + l_name + 4 > n || l_name == 0) return dfail(FTK_ERR_FORMAT, "truncated BAM reference list");
        std::string name((const char*)p + off, l_name - 1);
        off += l_name;
        int32_t l_ref = rd_i32(p + off);
        off += 4;
        if (!contig || name == contig) {
            ref_to_contig[r] = (int)t->contigs.size();
            t->contigs.push_back(Contig{name, l_ref, {}});
        }
    }
    // records (io/alignment.py:242-268)
    while (off + 4 <= n) {
        uint32_t bs = rd_u32(p + off);
        off += 4;
        if (bs < 32 || off + bs > n) return dfail(FTK_ERR_FORMAT, "truncated BAM record");
        const uint8_t* r = p + off;
        off += bs;
        int32_t ref_id = rd_i32(r);
        if (ref_id < 0 || (uint32_t)ref_id >= n_ref || ref_to_contig[ref_id] < 0) continue;
        if (32 + (size_t)r[8] + 4 * (size_t)rd_u16(r + 12) > bs) return dfail(FTK_ERR_FORMAT, "corrupt BAM record");
        Columns& c = t->contigs[ref_to_contig[ref_id]].c;
        bam_record(r, bs, c);  // io/alignment.py:60-71,242-268 (ftk_bamrule.h)
    }
    for (auto& ct : t->contigs) { t->skipped[0] += ct.c.skipped[0]; t->skipped[1] += ct.c.skipped[1]; }
    sw.lap("records");
    // the kernels need start-sorted fragments; read1 order is by read position (kept in `ord`)
    for (auto& ct : t->contigs) sort_by_start(ct.c, n_threads);
    sw.lap("sort by start");
    for (auto& ct : t->contigs) {
        pack(ct);
        if (!ct.p.base) return dfail(FTK_ERR_OOM, "out of host memory");
    }
    sw.lap("pack");
    *out = t.release();
    return FTK_OK;
}

// The C boundary never lets a C++ exception through: running out of host memory is an error code.
int ftk_fragfile_decode(const char* path, const char* contig, int n_threads, ftk_fragtable** out) {
    try {
        return ftk_fragfile_decode_impl(path, contig, n_threads, out);
    } catch (const std::exception& ex) {
        if (out) *out = nullptr;
        return dfail(FTK_ERR_OOM, ex.what());
    }
}

int ftk_bam_decode(const char* path, const char* contig, int n_threads, ftk_fragtable** out) {
    try {
        return ftk_bam_decode_impl(path, contig, n_threads, out);
    } catch (const std::exception& ex) {
        if (out) *out = nullptr;
        return dfail(FTK_ERR_OOM, ex.what());
    }
}

int ftk_fragtable_is_bed6(const ftk_fragtable* t) { return t && t->bed6 ? 1 : 0; }
int ftk_fragtable_n_contigs(const ftk_fragtable* t) { return t ? (int)t->contigs.size() : 0; }
const char* ftk_fragtable_contig_name(const ftk_fragtable* t, int i) {
    return (t && i >= 0 && i < (int)t->contigs.size()) ? t->contigs[i].name.c_str() : nullptr;
}
int64_t ftk_fragtable_contig_length(const ftk_fragtable* t, int i) {
    return (t && i >= 0 && i < (int)t->contigs.size()) ? t->contigs[i].length : -1;
}
int64_t ftk_fragtable_contig_rows(const ftk_fragtable* t, int i) {
    return (t && i >= 0 && i < (int)t->contigs.size()) ? (int64_t)t->contigs[i].p.rows : -1;
}
int ftk_fragtable_columns(const ftk_fragtable* t, int i, const int32_t** start, const int32_t** end,
                          const uint8_t** mapq, const uint8_t** strand, const int32_t** r1_start,
                          const int32_t** r1_end) {
    if (!t || i < 0 || i >= (int)t->contigs.size()) return dfail(FTK_ERR_NO_CONTIG, "contig index %d out of range", i);
    const Packed& c = t->contigs[i].p;
    if (start) *start = c.start;
    if (end) *end = c.end;
    if (mapq) *mapq = c.mapq;
    if (strand) *strand = c.strand;
    if (r1_start) *r1_start = c.r1s;
    if (r1_end) *r1_end = c.r1e;
    return FTK_OK;
}
int ftk_fragtable_order(const ftk_fragtable* t, int i, const int32_t** order) {
    if (!t || i < 0 || i >= (int)t->contigs.size() || !order) return dfail(FTK_ERR_NO_CONTIG, "contig index %d out of range", i);
    *order = t->contigs[i].p.ord;
    return FTK_OK;
}
int ftk_fragtable_is_pinned(const ftk_fragtable* t, int i) {
    return (t && i >= 0 && i < (int)t->contigs.size() && t->contigs[i].p.pinned) ? 1 : 0;
}
void ftk_fragtable_free(ftk_fragtable* t) { delete t; }

int ftk_fragtable_is_device(const ftk_fragtable* t, int i) {
    return (t && i >= 0 && i < (int)t->contigs.size() && t->contigs[i].dev) ? 1 : 0;
}

void* ftk_fragtable_ready_event(const ftk_fragtable* t, int i) {
    return ftk_fragtable_is_device(t, i) ? (void*)t->contigs[i].dev->ready : nullptr;
}

int ftk_fragtable_read1_to_host(const ftk_fragtable* t, int i, int32_t* r1_start, int32_t* r1_end, int32_t* order) {
    if (!t || i < 0 || i >= (int)t->contigs.size()) return dfail(FTK_ERR_NO_CONTIG, "contig index %d out of range", i);
    const Contig& ct = t->contigs[i];
    const size_t n = ct.p.rows;
    if (!ct.p.r1s || !ct.p.r1e) return dfail(FTK_ERR_INVALID, "the table holds no read1 columns (not a BAM table)");
    if (!ct.dev) {
        if (r1_start && n) memcpy(r1_start, ct.p.r1s, n * 4);
        if (r1_end && n) memcpy(r1_end, ct.p.r1e, n * 4);
        if (order && n) {
            if (!ct.p.ord) return dfail(FTK_ERR_INVALID, "the table holds no order column");
            memcpy(order, ct.p.ord, n * 4);
        }
        return FTK_OK;
    }
    const DevColumns& d = *ct.dev;
    bool ok = hipSetDevice(d.device) == hipSuccess && hipEventSynchronize(d.ready) == hipSuccess;
    if (ok && r1_start && n) ok = hipMemcpy(r1_start, d.r1s, n * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok && r1_end && n) ok = hipMemcpy(r1_end, d.r1e, n * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok && order && n) ok = d.ord && hipMemcpy(order, d.ord, n * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        return dfail(FTK_ERR_HIP, "cannot copy the read1 columns back");
    }
    return FTK_OK;
}

int ftk_fragtable_columns_to_host(const ftk_fragtable* t, int i, int32_t* start, int32_t* end, uint8_t* mapq,
                                  uint8_t* strand) {
    if (!t || i < 0 || i >= (int)t->contigs.size()) return dfail(FTK_ERR_NO_CONTIG, "contig index %d out of range", i);
    const Contig& ct = t->contigs[i];
    const size_t n = ct.p.rows;
    if (!ct.dev) {
        if (start && n) memcpy(start, ct.p.start, n * 4);
        if (end && n) memcpy(end, ct.p.end, n * 4);
        if (mapq && n) memcpy(mapq, ct.p.mapq, n);
        if (strand && n) memcpy(strand, ct.p.strand, n);
        return FTK_OK;
    }
    const DevColumns& d = *ct.dev;
    bool ok = hipSetDevice(d.device) == hipSuccess && hipEventSynchronize(d.ready) == hipSuccess;
    if (ok && start && n) ok = hipMemcpy(start, d.start, n * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok && end && n) ok = hipMemcpy(end, d.end, n * 4, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok && mapq && n) ok = hipMemcpy(mapq, d.mapq, n, hipMemcpyDeviceToHost) == hipSuccess;
    if (ok && strand && n) ok = hipMemcpy(strand, d.strand, n, hipMemcpyDeviceToHost) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        return dfail(FTK_ERR_HIP, "cannot copy the device columns to the host");
    }
    return FTK_OK;
}

int ftk_host_alloc(int64_t bytes, void** out) {
    if (!out || bytes < 0) return dfail(FTK_ERR_INVALID, "ftk_host_alloc: bad argument");
    *out = nullptr;
    if (!have_hip_device()) return dfail(FTK_ERR_NO_DEVICE, "ftk_host_alloc: no HIP device (page-locked memory needs the driver)");
    void* p = result_cache().alloc((size_t)std::max<int64_t>(bytes, 1));
    if (!p) return dfail(FTK_ERR_OOM, "ftk_host_alloc: cannot page-lock %lld more bytes (driver refusal or the limit on page-locked results)", (long long)bytes);
    *out = p;
    return FTK_OK;
}

int ftk_host_alloc_pageable(int64_t bytes, void** out) {
    if (!out || bytes < 0) return dfail(FTK_ERR_INVALID, "ftk_host_alloc_pageable: bad argument");
    *out = plain_result_cache().alloc((size_t)std::max<int64_t>(bytes, 1));
    if (!*out) return dfail(FTK_ERR_OOM, "ftk_host_alloc_pageable: cannot allocate %lld bytes", (long long)bytes);
    return FTK_OK;
}

void ftk_host_free(void* p) {
    if (p && !result_cache().release(p)) (void)plain_result_cache().release(p);  // a pointer that is not one of ours is ignored
}

}  // extern "C"

// ---------------------------------------------------------------------------------------
// Streaming decoder: one contig at a time, decoded ahead of the consumer
// ---------------------------------------------------------------------------------------
// A producer thread reads the file in pieces of whole BGZF blocks, inflates and parses each
// piece on the worker threads and hands every finished contig (coordinate-sorted files keep a
// contig's rows together) to a bounded queue as a one-contig ftk_fragtable in page-locked
// memory.  The consumer uploads / computes on contig k while contig k+1 is being decoded, and
// host memory stays bounded by the piece size plus `max_queued` contigs however large the file.
namespace {

// compressed bytes read per piece (FTK_STREAM_PIECE overrides it: the tests use small pieces to
// exercise blocks, lines and records that straddle a piece boundary)
const size_t kStreamPiece = [] {
    const char* e = getenv("FTK_STREAM_PIECE");
    const long long v = e ? atoll(e) : 0;
    // (at most 120 MB: a large BAM doubles it, and the device inflate addresses a piece by 32-bit bit positions - 2^28 bytes a launch)
    return v >= (1 << 16) ? (size_t)std::min<long long>(v, 120ll << 20) : (size_t)(48u << 20);
}();

// bytes of inflated BAM per speculative stretch of the record chain (FTK_BAM_STRETCH: the tests make the
// stretches tiny so that small files exercise the guess / check / redo logic)
const size_t kBamStretch = [] {
    const char* e = getenv("FTK_BAM_STRETCH");
    const long long v = e ? atoll(e) : 0;
    return v >= 64 ? (size_t)v : (size_t)(1u << 20);
}();

struct BamRun {
    int ref = -1;
    Columns c;
};

// Growable byte buffer without zero-fill (a std::vector would memset every piece it grows by).
// `pinned`: page-locked memory from the library's recycled blocks instead - the pieces of a BAM stream whose records
// are parsed on the device go up straight from the buffer they were read into.
struct RawBuf {
    uint8_t* p = nullptr;
    size_t cap = 0;
    size_t head = 0;  // data() starts here (the read-ahead piece leaves room in front for carried bytes)
    bool pinned = false;
    RawBuf() = default;
    RawBuf(const RawBuf&) = delete;
    RawBuf& operator=(const RawBuf&) = delete;
    ~RawBuf() {
        if (!p) return;
        if (pinned) pinned_free(p); else huge_unmap(p, cap);
    }
    bool reserve(size_t n) {
        if (n <= cap) return true;
        if (pinned) {
            const size_t want = std::max(n, cap + cap / 2);
            uint8_t* q = (uint8_t*)pinned_alloc(want);
            if (!q) return false;
            if (p) {
                memcpy(q, p, cap);
                pinned_free(p);
            }
            p = q;
            cap = want;
            return true;
        }
        const size_t want = huge_round(std::max(n, cap + cap / 2));
        uint8_t* q = p ? huge_remap(p, cap, want) : huge_map(want);
        if (!q) return false;
        p = q;
        touch_pages(p, cap, want);
        cap = want;
        return true;
    }
    uint8_t* data() { return p + head; }
    void swap(RawBuf& o) {
        std::swap(p, o.p);
        std::swap(cap, o.cap);
        std::swap(head, o.head);
        std::swap(pinned, o.pinned);
    }
};

// Pack a contig held as a list of runs: one block (page-locked when a device is present), the runs
// copied to their final places by `n_threads` threads.
void pack_parts(Contig& ct, int n_threads) {
    size_t m = 0;
    std::vector<size_t> at;
    for (auto& c : ct.parts) { at.push_back(m); m += c.start.size(); }
    const size_t total = packed_bytes(m, false);
    Packed& p = ct.p;
    if (have_hip_device() && (p.base = pinned_alloc(total)) != nullptr) {
        p.pinned = true;
    } else {
        p.base = malloc(total);
        p.pinned = false;
    }
    if (!p.base) return;
    place(p, (char*)p.base, m, false);
    std::atomic<size_t> next{0};
    auto work = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= ct.parts.size()) break;
            const Columns& c = ct.parts[k];
            const size_t n = c.start.size(), o = at[k];
            if (!n) continue;
            memcpy(p.start + o, c.start.data(), n * 4);
            memcpy(p.end + o, c.end.data(), n * 4);
            memcpy(p.mapq + o, c.mapq.data(), n);
            memcpy(p.strand + o, c.strand.data(), n);
        }
    };
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, ct.parts.size()));
    parallel_run(nt, [&](int) { work(); });
    std::vector<Columns>().swap(ct.parts);
}

// A BAM contig held as runs in read1-position order -> its page-locked block in fragment-start order:
// keys (start << 32 | file rank) built, sorted and merged in parallel, then every column gathered from
// the runs straight into its final place - no concatenated or sorted intermediate copy of the contig.
void pack_bam_parts(Contig& ct, int n_threads) {
    Stopwatch sw;
    std::vector<size_t> at;
    size_t m = 0;
    for (auto& c : ct.parts) { at.push_back(m); m += c.start.size(); }
    at.push_back(m);
    const size_t total = packed_bytes(m, true);
    Packed& p = ct.p;
    if (have_hip_device() && (p.base = pinned_alloc(total)) != nullptr) {
        p.pinned = true;
    } else {
        p.base = malloc(total);
        p.pinned = false;
    }
    if (!p.base) return;
    place(p, (char*)p.base, m, true);
    sw.lap("  packer: block");
    const size_t np = ct.parts.size();
    if (m) {
        std::unique_ptr<uint64_t[]> key(new uint64_t[m]);  // (not zeroed: the threads below write every key)
        std::atomic<int> unsorted{0};
        {
            std::atomic<size_t> next{0};
            parallel_run((int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, np)), [&](int) {
                bool bad = false;
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= np) break;
                    const std::vector<int32_t>& st = ct.parts[k].start;
                    for (size_t i = 0; i < st.size(); ++i) {
                        key[at[k] + i] = ((uint64_t)(uint32_t)st[i] << 32) | (uint64_t)(at[k] + i);
                        bad |= i > 0 && st[i] < st[i - 1];
                    }
                    // (across parts: the first start of this one against the last of the one before)
                    if (k > 0 && !st.empty() && !ct.parts[k - 1].start.empty()) bad |= st.front() < ct.parts[k - 1].start.back();
                }
                if (bad) unsorted.store(1);
            });
        }
        int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)std::max(n_threads, 1), m / 65536));
        sw.lap("  packer: keys");
        if (unsorted.load()) nt = sort_keys(key.get(), m, n_threads);
        sw.lap("  packer: sort");
        parallel_run(nt, [&](int t) {
            size_t k = 0;
            for (size_t i = m * (size_t)t / (size_t)nt, e = m * (size_t)(t + 1) / (size_t)nt; i < e; ++i) {
                const size_t j = (size_t)(uint32_t)key[i];
                while (j < at[k]) --k;            // nearly sorted: the run changes rarely
                while (j >= at[k + 1]) ++k;
                const Columns& c = ct.parts[k];
                const size_t o = j - at[k];
                p.start[i] = c.start[o];
                p.end[i] = c.end[o];
                p.mapq[i] = c.mapq[o];
                p.strand[i] = c.strand[o];
                p.r1s[i] = c.r1s[o];
                p.r1e[i] = c.r1e[o];
                p.ord[i] = (int32_t)j;
            }
        });
    }
    sw.lap("  packer: gather");
    std::vector<Columns>().swap(ct.parts);
    sw.lap("  packer: free parts");
}

// Complete text lines [b, e) -> runs in file order (segments parsed in parallel).
void parse_text_parallel(const char* b, const char* e, bool bed6, const char* only, int n_threads,
                         std::vector<Run>* out) {
    const size_t n = (size_t)(e - b);
    int nseg = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, n / (1 << 16) + 1));
    std::vector<const char*> cut(nseg + 1);
    cut[0] = b;
    cut[nseg] = e;
    for (int i = 1; i < nseg; ++i) {
        const char* q = b + n * (size_t)i / (size_t)nseg;
        if (q < cut[i - 1]) q = cut[i - 1];
        const char* nl = (const char*)memchr(q, '\n', (size_t)(e - q));
        cut[i] = nl ? nl + 1 : e;
    }
    std::vector<std::vector<Run>> seg(nseg);
    parallel_run(nseg, [&](int i) { parse_segment(cut[i], cut[i + 1], bed6, only, &seg[i]); });
    for (auto& v : seg)
        for (auto& r : v) out->push_back(std::move(r));
}

// Does p[o..) look like the start of a BAM alignment record?  Only used to GUESS where a thread may
// enter the record chain in the middle of a piece (run_bam); a wrong guess is detected and redone.
inline bool plausible_record(const uint8_t* p, size_t o, size_t m, int n_ref) {
    if (o + 36 > m) return false;
    const uint32_t bs = rd_u32(p + o);
    if (bs < 32 || bs > (1u << 24)) return false;
    const uint8_t* r = p + o + 4;
    const int32_t ref = rd_i32(r), pos = rd_i32(r + 4), next_ref = rd_i32(r + 20), next_pos = rd_i32(r + 24);
    if (ref < -1 || ref >= n_ref || next_ref < -1 || next_ref >= n_ref || pos < -1 || next_pos < -1) return false;
    const uint32_t l_name = r[8], n_cigar = rd_u16(r + 12);
    const int32_t l_seq = rd_i32(r + 16);
    if (l_name < 2 || l_seq < 0) return false;  // (a missing name is "*": never empty)
    if (rd_u16(r + 14) & 0xf000) return false;   // undefined flag bits
    const uint64_t need = 32 + (uint64_t)l_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + (uint64_t)l_seq;
    if (need > bs) return false;
    if (o + 36 + l_name <= m) {  // the read name: printable characters, NUL-terminated
        if (r[32 + l_name - 1] != 0) return false;
        for (uint32_t k = 0; k + 1 < l_name; ++k)
            if (r[32 + k] < 33 || r[32 + k] > 126) return false;
    }
    return true;
}

inline size_t guess_record_start(const uint8_t* p, size_t from, size_t m, int n_ref) {
    // A candidate whose very first link leaves the piece proves nothing - and two bytes in front of a real record
    // of reference 0 there is one: the low half of the real block_size lands in the high half of a 32-bit size of
    // megabytes, the zeros behind it read as reference 0 (11 % of the stretches of a 60x file were redone for
    // that).  Such a candidate is kept only as a fallback for when nothing checkable follows.
    size_t fallback = SIZE_MAX;
    for (size_t o = from; o + 36 <= m; ++o) {
        if (!plausible_record(p, o, m, n_ref)) continue;
        size_t o2 = o + 4 + (size_t)rd_u32(p + o);
        bool ok = true;
        int checked = 0;
        for (int k = 0; k < 2; ++k) {  // two more links must hold, unless the piece ends first
            if (o2 + 36 > m) break;
            if (!plausible_record(p, o2, m, n_ref)) { ok = false; break; }
            ++checked;
            o2 += 4 + (size_t)rd_u32(p + o2);
        }
        if (!ok) continue;
        if (checked) return o;
        if (fallback == SIZE_MAX) fallback = o;
        if (o - from > (size_t(1) << 16)) break;  // (do not scan a long unverifiable tail byte by byte)
    }
    return fallback;
}

}  // namespace

// ---- tabix (.tbi) / BAM (.bai) index: where a contig's rows start and end in the file --------
// Both hold, per reference, bins of chunks [virtual begin, virtual end) (virtual = block file
// offset << 16 | offset inside the inflated block).  htslib's pseudo-bin 37450 carries the span of the
// whole reference in its first chunk; without it the span is the hull of the chunks.
namespace {

struct IndexSpan {
    bool usable = false;   // the index could be read
    bool present = false;  // ... and lists the contig with at least one chunk
    uint64_t beg = 0, end = 0;
    // region look-ups (tabix): where the rows that overlap the region start, and where - by the 16 kb linear index -
    // the rows that start behind the region begin (a hint: a row longer than an index window can sit behind it)
    bool region = false;
    uint64_t reg_beg = 0, reg_soft_end = 0;
};

inline uint64_t rd_u64(const uint8_t* p) { return (uint64_t)rd_u32(p) | ((uint64_t)rd_u32(p + 4) << 32); }

// ref < 0: look the contig up by name (tabix); else by reference id (BAI)
IndexSpan index_lookup(const std::string& index_path, bool bai, const std::string& name, int ref, long long reg_start = -1,
                       long long reg_stop = -1) {
    IndexSpan out;
    // The index image (a tabix index inflated) of the file asked for last stays in memory: a whole-genome .tbi is
    // ~1.5 MB to read and inflate - 3 ms on one thread - and region reads ask for it once per region (a rank's two
    // partial contigs, every one-interval API call).  Keyed by path, size and modification time.
    struct Cached {
        std::string path;
        long long size = -1, mtime_ns = 0;
        std::shared_ptr<std::vector<uint8_t>> image;
    };
    static std::mutex cache_mu;
    static Cached cache[2];
    static unsigned cache_turn = 0;
    std::shared_ptr<std::vector<uint8_t>> image;
    struct stat ist;
    if (stat(index_path.c_str(), &ist) != 0) return out;
    const long long isize = (long long)ist.st_size,
                    imtime = (long long)ist.st_mtim.tv_sec * 1000000000LL + (long long)ist.st_mtim.tv_nsec;
    {
        std::lock_guard<std::mutex> lk(cache_mu);
        for (auto& c : cache)
            if (c.image && c.path == index_path && c.size == isize && c.mtime_ns == imtime) image = c.image;
    }
    if (!image) {
        Bytes raw, img;
        if (!read_file(index_path.c_str(), &raw) || raw.size() < 8) return out;
        const Bytes* src = &raw;
        if (!bai) {
            if (inflate_all(raw, 1, &img) != FTK_OK || img.size() < 36) return out;
            src = &img;
        }
        image = std::make_shared<std::vector<uint8_t>>(src->data(), src->data() + src->size());
        std::lock_guard<std::mutex> lk(cache_mu);
        Cached& slot = cache[cache_turn++ & 1u];
        slot.path = index_path;
        slot.size = isize;
        slot.mtime_ns = imtime;
        slot.image = image;
    }
    const uint8_t* p = image->data();
    const size_t n = image->size();
    if (n < 8 || memcmp(p, bai ? "BAI\1" : "TBI\1", 4) != 0) return out;
    if (!bai && n < 36) return out;
    const int32_t n_ref = rd_i32(p + 4);
    size_t o = 8;
    if (!bai) {
        const int32_t l_nm = rd_i32(p + 32);
        o = 36;
        if (l_nm < 0 || o + (size_t)l_nm > n) return out;
        ref = -1;
        int k = 0;
        for (size_t a = o; a < o + (size_t)l_nm && k < n_ref; ++k) {
            const char* nm = (const char*)p + a;
            const size_t len = strnlen(nm, o + l_nm - a);
            if (name.size() == len && memcmp(nm, name.data(), len) == 0) ref = k;
            a += len + 1;
        }
        o += (size_t)l_nm;
        out.usable = true;
        if (ref < 0) return out;  // the file has no such contig
    }
    if (n_ref < 0 || ref >= n_ref) { out.usable = bai; return out; }
    for (int r = 0; r <= ref; ++r) {
        if (o + 4 > n) return IndexSpan{};
        const int32_t n_bin = rd_i32(p + o);
        o += 4;
        uint64_t lo = UINT64_MAX, hi = 0;
        bool pseudo = false;
        for (int32_t b = 0; b < n_bin; ++b) {
            if (o + 8 > n) return IndexSpan{};
            const uint32_t bin = rd_u32(p + o);
            const int32_t n_chunk = rd_i32(p + o + 4);
            o += 8;
            if (n_chunk < 0 || o + 16 * (size_t)n_chunk > n) return IndexSpan{};
            if (r == ref) {
                if (bin == 37450 && n_chunk >= 1) {
                    lo = rd_u64(p + o);
                    hi = rd_u64(p + o + 8);
                    pseudo = true;
                } else if (!pseudo && bin != 37450) {
                    for (int32_t c = 0; c < n_chunk; ++c) {
                        lo = std::min(lo, rd_u64(p + o + 16 * (size_t)c));
                        hi = std::max(hi, rd_u64(p + o + 16 * (size_t)c + 8));
                    }
                }
            }
            o += 16 * (size_t)n_chunk;
        }
        if (o + 4 > n) return IndexSpan{};
        const int32_t n_intv = rd_i32(p + o);
        o += 4;
        if (n_intv < 0 || o + 8 * (size_t)n_intv > n) return IndexSpan{};
        const uint8_t* ioff = p + o;
        o += 8 * (size_t)n_intv;
        if (r == ref) {
            out.usable = true;
            if (lo != UINT64_MAX && hi > lo) { out.present = true; out.beg = lo; out.end = hi; }
            if (out.present && reg_start >= 0 && reg_stop > reg_start && n_intv > 0) {
                // linear index: ioff[w] = the smallest virtual offset of a row that overlaps [w * 16384, (w + 1) * 16384)
                // (0: none seen up to there).  Every row overlapping the region's first base overlaps its window, so
                // nothing before ioff[w0] is needed; rows that START in window w1 = (stop >> 14) + 1 or later lie
                // behind the region.
                const long long w0 = std::min<long long>(reg_start >> 14, n_intv - 1), w1 = (reg_stop >> 14) + 1;
                uint64_t b = rd_u64(ioff + 8 * (size_t)w0);
                out.reg_beg = b >= lo && b < hi ? b : lo;
                uint64_t e = w1 < n_intv ? rd_u64(ioff + 8 * (size_t)w1) : 0;
                out.reg_soft_end = e >= out.reg_beg && e < hi ? e : hi;  // (== : the region lies in a stretch without rows)
                out.region = true;
            }
        }
    }
    return out;
}

std::string index_path_of(const std::string& path, bool bam) {
    if (!bam) return path + ".tbi";
    std::string a = path + ".bai";
    if (FILE* f = fopen(a.c_str(), "rb")) { fclose(f); return a; }
    if (path.size() > 4) return path.substr(0, path.size() - 4) + ".bai";
    return a;
}

}  // namespace

struct ftk_fragstream {
    std::string path, only;
    bool has_only = false, bam = false, bed6 = false;
    // compressed bytes per piece of THIS stream: kStreamPiece, doubled for a large BAM parsed on the device (run_guarded)
    size_t piece_bytes = kStreamPiece;
    int n_threads = 1;
    size_t max_queued = 2;
    FILE* fp = nullptr;
    std::thread producer;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<ftk_fragtable*> ready;
    bool finished = false, stop = false;
    bool consumer_waiting = false;  // ftk_fragstream_next is blocked on an empty queue
    int err = FTK_OK;
    std::string errmsg;
    // BAM header
    std::vector<std::string> ref_names;
    std::vector<int64_t> ref_lens;
    bool header_ready = false;
    double stage_ms[6] = {0, 0, 0, 0, 0, 0};  // read, inflate, parse, merge, emit, other: set when the producer is done
    // BAM records the reference handles differently from "no fragment" (ftk_bamrule.h), met so far: [0] fragments the
    // columns cannot hold (negative start / beyond int32), [1] CIGAR-less read1 with TLEN < 0 (ftk_fragstream_skipped)
    std::atomic<int64_t> skipped[2] = {{0}, {0}};

    // Text files on a stream opened with ftk_fragstream_open_on: the rows are parsed on this GPU
    // (run_text_device) and the tables handed out hold device columns.
    int device = -1;
    int inflate_device = -1;  // BAM streams opened with a device: where run_bam inflates the pieces
    hipStream_t pstream = nullptr;
    bool emit_device(Contig&& ct);
    bool run_text_device(RawBuf& first, size_t first_n);
    // A row / comment line longer than the device carry (kTextCarryMax) cannot be moved from piece to piece on the
    // device: run_text_device then asks for a second pass over the file with the inflate on the host threads (whose
    // carry has no limit); the contigs already handed out are skipped in that pass.
    bool want_host_restart = false, host_inflate_only = false;
    std::set<std::string> emitted_names;

    // Hand one finished contig to the consumer.  Sorting (BAM), packing into page-locked memory and
    // waiting for queue space happen on a helper thread, one contig at a time (so the order is kept),
    // while the producer already decodes the next contig's pieces.
    std::thread packer;
    std::atomic<int> packer_ok{1};
    bool emit(Contig&& ct) {
        if (packer.joinable()) packer.join();
        if (!packer_ok.load()) return false;
        std::shared_ptr<Contig> held(new Contig(std::move(ct)));
        packer = std::thread([this, held] {
            std::unique_ptr<ftk_fragtable> t(new ftk_fragtable());
            t->bam = bam;
            t->bed6 = bed6;
            t->contigs.push_back(std::move(*held));
            Contig& c = t->contigs[0];
            Stopwatch sw;
            if (!c.parts.empty()) {
                if (bam) pack_bam_parts(c, n_threads);
                else pack_parts(c, n_threads);
                sw.lap(bam ? "packer: sort + gather runs" : "packer: pack runs");
            } else {
                if (bam) sort_by_start(c.c, n_threads);
                sw.lap("packer: sort by start");
                pack(c);
                sw.lap("packer: pack");
            }
            if (!c.p.base) { fail(FTK_ERR_OOM, "out of host memory"); packer_ok = 0; return; }
            std::unique_lock<std::mutex> lk(mu);
            cv.wait(lk, [&] { return stop || ready.size() < max_queued; });
            if (stop) { packer_ok = 0; return; }
            ready.push_back(t.release());
            cv.notify_all();
        });
        return true;
    }
    bool flush() {  // wait for the last contig to be queued
        if (packer.joinable()) packer.join();
        return packer_ok.load() != 0;
    }
    bool fail(int code, const char* msg) {
        std::lock_guard<std::mutex> lk(mu);
        if (err == FTK_OK) { err = code; errmsg = msg; }
        return false;
    }
    void run();          // producer thread body: run_guarded() with exceptions turned into a stream error
    void run_guarded();
    bool run_text(RawBuf& first, size_t first_n);
    bool run_bam(RawBuf& first, size_t first_n);
    // BAM with the records parsed ON THE DEVICE (ftk_bamparse.hip): the inflated bytes stay in HBM, the tables handed
    // out hold device columns sorted by fragment start.  A piece whose record chain the device cannot settle (or a
    // header larger than a piece) makes the stream start over on the host path (run_bam), which skips the contigs
    // already handed out (emitted_refs).
    bool run_bam_device(RawBuf& first, size_t first_n);
    bool emit_device_bam(Contig&& ct);
    std::set<int> emitted_refs;
    // pieces a whole-file read will come to (0 for an index-driven read of a contig or region: short, and its length is
    // not the file's)
    int pieces_expected() const {
        struct stat sb;
        if (read_end >= 0 || !fp || fstat(fileno(fp), &sb) != 0 || !S_ISREG(sb.st_mode) || piece_bytes == 0) return 0;
        return (int)std::min<long long>((long long)sb.st_size / (long long)piece_bytes + 1, 1 << 20);
    }
    // single-contig requests with a usable index: read only the file range holding the contig
    long long read_end = -1;      // file offset to stop reading at (-1: none)
    bool partial_tail_ok = false;  // the range may end inside a block that belongs to the next contig
    size_t first_skip = 0;        // bytes of the first inflated block that precede the contig
    long long first_piece_off = -1;  // file offset of the first piece handed to run_* (-1: unknown)
    // Region reads (ftk_fragstream_open_region, text files with a tabix index): the stream hands out the contig's rows
    // from the first that can overlap [reg_start, reg_stop) to the last that starts before reg_stop - a superset is
    // allowed, nothing of the region may be missing.  The read first stops at the linear index's hint
    // (read_end < hard_read_end); the device parser's rows then say whether the region is complete (a row starting at
    // or behind reg_stop, or another contig's rows, were seen) and the read goes on in steps if not.
    bool has_region = false;
    long long reg_start = 0, reg_stop = 0;
    long long hard_read_end = -1;  // where the contig's rows end (read_end of a whole-contig read)
    bool range_limited = false;    // the last read_piece() came back short because of read_end, not the file's end
    // One piece of the file -> dst; returns the bytes read (short at the end of the file / of the range).
    // FTK_STREAM_RAMP=<bytes>: the first reads of a stream short - that many bytes, then twice as much each time up to
    // piece_bytes.  The idea: a launch of the inflate kernel lasts one block's chain whatever its size, so the first
    // rows would reach HBM after the time it takes to read and send up 4 MB instead of 48 (96 for a large BAM).
    // Measured (tools/env_ab.sh, alternating runs on one box and file): the whole genome 0.123 -> 0.128 s, the 5.9 GB
    // BAM 0.265 -> 0.272 s - the short launches fill the chip worse than the wait they save; OFF by default.  `base`
    // is what a read asks for before any range limit; fewer bytes than that = the file, or the range, ended there
    // (last_want, set by fill()).
    size_t reads_issued = 0, last_want = 0, ahead_want = 0;
    size_t next_want() {
        static const size_t ramp0 = [] {
            const char* e = getenv("FTK_STREAM_RAMP");
            return e ? (size_t)std::max(0ll, atoll(e)) : (size_t)0;
        }();
        size_t w = piece_bytes;
        if (ramp0 && reads_issued < 6) w = std::min(piece_bytes, ramp0 << reads_issued);
        ++reads_issued;
        return w;
    }
    size_t read_piece(uint8_t* dst, size_t base) {
        size_t want = base;
        range_limited = false;
        if (read_end >= 0) {
            const long long pos = ftell(fp);
            want = pos >= read_end ? 0 : (size_t)std::min<long long>((long long)base, read_end - pos);
            range_limited = want < base;
        }
        size_t got = 0;
        bool done = false;
        static const int n_readers = [] {  // FTK_READ_THREADS=<1..8>
            const char* e = getenv("FTK_READ_THREADS");
            return e ? std::max(1, std::min(8, atoi(e))) : 4;
        }();
        if (want >= (size_t(8) << 20) && n_readers > 1) {
            // a large piece of a regular file: four threads pread their quarters (one thread copies ~6 GB/s out of
            // the page cache - 8 ms per 48 MB piece, as long as the GPU takes to inflate it)
            const long long pos = ftell(fp);
            struct stat sb;
            if (pos >= 0 && fstat(fileno(fp), &sb) == 0 && S_ISREG(sb.st_mode) && (long long)sb.st_size > pos) {
                const size_t n = (size_t)std::min<long long>((long long)want, (long long)sb.st_size - pos);
                const int kReaders = n_readers;
                std::atomic<int> failed{0};
                auto part = [&](int t) {
                    size_t a = n * (size_t)t / kReaders;
                    const size_t b = n * (size_t)(t + 1) / kReaders;
                    while (a < b) {
                        const ssize_t r = pread(fileno(fp), dst + a, b - a, (off_t)(pos + (long long)a));
                        if (r <= 0) { failed.store(1); return; }
                        a += (size_t)r;
                    }
                };
                std::vector<std::thread> helpers;
                for (int t = 1; t < kReaders; ++t) helpers.emplace_back(part, t);
                part(0);
                for (auto& h : helpers) h.join();
                if (!failed.load() && fseek(fp, (long)(pos + (long long)n), SEEK_SET) == 0) {
                    got = n;
                    done = true;
                } else if (fseek(fp, (long)pos, SEEK_SET) != 0) {
                    return 0;
                }
            }
        }
        if (!done) got = want ? fread(dst, 1, want, fp) : 0;
        if (got == want && want) {
            // ask the kernel for the piece after this one (a hint only; failure is ignored)
            const long long pos = ftell(fp);
            if (pos >= 0) (void)posix_fadvise(fileno(fp), (off_t)pos, (off_t)piece_bytes, POSIX_FADV_WILLNEED);
        }
        return got;
    }
    // Read-ahead: while a piece is inflated and parsed, a helper thread reads the next one into `ahead`,
    // kHead bytes in: the bytes carried over (less than one BGZF block) are put in front of it and the two
    // buffers trade places - the file read (12 GB/s from the page cache, far less from a cold disk)
    // leaves the producer's critical path.  Only the helper touches `fp` while a read is in flight.
    static constexpr size_t kHead = size_t(1) << 16;
    RawBuf ahead;
    std::future<size_t> ahead_got;
    bool ahead_ok = true;  // false while the BAM header is probed before an index seek (the read would be thrown away)
    void start_ahead() {
        if (!ahead_ok || !ahead.reserve(kHead + piece_bytes)) return;
        ahead.head = 0;
        ahead_want = next_want();
        ahead_got = std::async(std::launch::async, [this] { return read_piece(ahead.p + kHead, ahead_want); });
    }
    void drain_ahead() {  // before anything else moves the file position
        if (ahead_got.valid()) (void)ahead_got.get();
    }
    // A text stream sends its GPU pieces up straight from the (page-locked) buffer they were read into.  The copy is
    // asynchronous: the caller hands fill() the event recorded behind it (buf_in_flight), the buffer rests until the
    // event is done, and the next read goes into one that has rested (up to three rest, so the producer never waits).
    struct Resting {
        std::unique_ptr<RawBuf> b;
        hipEvent_t ev;
    };
    std::deque<Resting> resting;
    std::vector<hipEvent_t> up_events;  // idle events of rested buffers
    hipEvent_t buf_in_flight = nullptr;
    hipEvent_t take_up_event() {
        if (!up_events.empty()) {
            hipEvent_t e = up_events.back();
            up_events.pop_back();
            return e;
        }
        hipEvent_t e = nullptr;
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return e;
    }
    void drop_resting() {
        for (auto& r : resting) {
            if (hipEventSynchronize(r.ev) != hipSuccess) (void)hipGetLastError();
            up_events.push_back(r.ev);
        }
        resting.clear();
        if (buf_in_flight) {
            if (hipEventSynchronize(buf_in_flight) != hipSuccess) (void)hipGetLastError();
            up_events.push_back(buf_in_flight);
        }
        buf_in_flight = nullptr;
        for (hipEvent_t e : up_events) (void)hipEventDestroy(e);
        up_events.clear();
    }
    // read the next piece after the `carry` bytes at the front of buf (at carry_off of buf when a copy of buf is in
    // flight and the caller could not move them); returns bytes now in buf
    size_t fill(RawBuf& buf, size_t carry, size_t carry_off = 0) {
        size_t got;
        if (buf_in_flight && !ahead_got.valid()) {  // no read-ahead to trade places with: wait for the copy
            if (hipEventSynchronize(buf_in_flight) != hipSuccess) (void)hipGetLastError();
            up_events.push_back(buf_in_flight);
            buf_in_flight = nullptr;
        }
        if (ahead_got.valid()) {
            got = ahead_got.get();
            last_want = ahead_want;
            if (carry <= kHead) {
                if (carry) memcpy(ahead.p + kHead - carry, buf.data() + carry_off, carry);
                ahead.head = kHead - carry;
            } else {  // (not with BGZF blocks, which are at most 64 KB)
                if (!ahead.reserve(carry + piece_bytes)) return carry;
                memmove(ahead.p + carry, ahead.p + kHead, got);
                memcpy(ahead.p, buf.data() + carry_off, carry);
                ahead.head = 0;
            }
            buf.swap(ahead);
            if (buf_in_flight) {  // the old buffer rests; the next read goes into one that has rested, or a new one
                resting.push_back({std::unique_ptr<RawBuf>(new RawBuf()), buf_in_flight});
                resting.back().b->swap(ahead);
                buf_in_flight = nullptr;
                ahead.pinned = true;
                if (resting.size() > 3 || hipEventQuery(resting.front().ev) == hipSuccess) {
                    if (hipEventSynchronize(resting.front().ev) != hipSuccess) (void)hipGetLastError();
                    ahead.swap(*resting.front().b);
                    up_events.push_back(resting.front().ev);
                    resting.pop_front();
                } else {
                    (void)hipGetLastError();  // (not ready is no error)
                }
            }
        } else {
            if (carry && carry_off) memmove(buf.data(), buf.data() + carry_off, carry);
            if (!buf.reserve(buf.head + carry + piece_bytes)) return carry;
            last_want = next_want();
            got = read_piece(buf.data() + carry, last_want);
        }
        if (got == last_want) start_ahead();
        return carry + got;
    }
    // seek to a contig's rows; false = index unusable (caller scans the whole file)
    bool seek_to(const IndexSpan& sp) {
        drain_ahead();
        if (fseek(fp, (long)(sp.beg >> 16), SEEK_SET) != 0) return false;
        first_skip = (size_t)(sp.beg & 0xffff);
        read_end = (long long)(sp.end >> 16) + 0x10000 + 64;  // through the block that holds the last row
        hard_read_end = read_end;
        partial_tail_ok = true;
        return true;
    }
    // complete BGZF blocks at the front of buf[0, n): block list + bytes consumed; false on corruption
    bool whole_blocks(const uint8_t* p, size_t n, bool eof, std::vector<Block>* blocks, size_t* used, size_t* total) {
        blocks->clear();
        size_t off = 0, tot = 0;
        while (off < n) {
            size_t bs = 0;
            if (off + 18 > n) break;
            const size_t q = gzip_header(p, n, off, &bs);
            if (q && bs && off + bs <= n) {
                if (q + 8 > off + bs) return false;
                const uint8_t* tr = p + off + bs - 8;
                const size_t isize = (size_t)tr[4] | ((size_t)tr[5] << 8) | ((size_t)tr[6] << 16) | ((size_t)tr[7] << 24);
                if (isize > (size_t(1) << 16)) return false;  // a BGZF block holds at most 64 KB of data (htslib refuses more too)
                blocks->push_back({q, off + bs - 8 - q, tot, isize});
                tot += isize;
                off += bs;
                continue;
            }
            if (p[off] != 31 || p[off + 1] != 139) return false;  // not at a block boundary
            if (q && bs && off + bs > n) break;                     // block continues in the next piece
            if (!q && n - off < (1u << 16)) break;                  // header itself is cut
            return false;
        }
        if (eof && off != n && !partial_tail_ok) return false;  // trailing garbage / truncated last block
        *used = off;
        *total = tot;
        return true;
    }
};

void ftk_fragstream::run_guarded() {
    RawBuf buf;
    ahead_ok = !(bam && has_only);  // run_bam decides about the index seek after the header
    {
        // a region is read as one only where the device inflates and parses the rows (run_text_device tells from the
        // parsed rows whether the region is complete); elsewhere the stream hands out the whole contig - a superset
        static const bool dev_inf = !(getenv("FTK_DEVICE_INFLATE") && atoi(getenv("FTK_DEVICE_INFLATE")) == 0);
        static const bool dev_bam_rec = !(getenv("FTK_DEVICE_BAM_PARSE") && atoi(getenv("FTK_DEVICE_BAM_PARSE")) == 0);
        if (!dev_inf || !has_only || (bam ? (inflate_device < 0 || !dev_bam_rec) : device < 0)) has_region = false;
    }
    if (has_only && !bam) {  // tabix index: jump straight to the contig's rows
        IndexSpan sp = index_lookup(index_path_of(path, false), false, only, -1, has_region ? reg_start : -1, has_region ? reg_stop : -1);
        if (has_region && !(sp.usable && sp.present && sp.region)) has_region = false;  // (the whole contig: a superset)
        if (sp.usable && !sp.present) {  // the file has no row of this contig
            std::lock_guard<std::mutex> lk(mu);
            finished = true;
            header_ready = true;
            cv.notify_all();
            return;
        }
        if (sp.usable && has_region) {
            // start at the region's first row; stop - for now - where the linear index says the rows behind it begin
            sp.beg = sp.reg_beg;
            if (!seek_to(sp)) { read_end = -1; partial_tail_ok = false; first_skip = 0; has_region = false; rewind(fp); }
            else read_end = std::min(hard_read_end, (long long)(sp.reg_soft_end >> 16) + 0x10000 + 64);
        } else if (sp.usable && !seek_to(sp)) { read_end = -1; partial_tail_ok = false; first_skip = 0; rewind(fp); }
    }
    {
        static const bool dev_bam = !(getenv("FTK_DEVICE_BAM_PARSE") && atoi(getenv("FTK_DEVICE_BAM_PARSE")) == 0) &&
                                    !(getenv("FTK_DEVICE_INFLATE") && atoi(getenv("FTK_DEVICE_INFLATE")) == 0);
        if (bam && inflate_device >= 0 && dev_bam && have_hip_device()) {
            buf.pinned = ahead.pinned = true;  // (see RawBuf)
            // A launch of the inflate kernel lasts one block's chain (~4 ms for BAM records) whatever its size, until
            // the chip's 5 120 wavefront slots are full; a 48 MB piece of BAM is ~1 900 blocks, and the stream keeps
            // about two such launches side by side: a third of the slots idle.  Pieces of 96 MB: a chr1-sized 60x
            // BAM 0.266 -> 0.233 s until resident (144 MB: no further gain; text streams, ~3 000 blocks per piece
            // and three fronts side by side, LOSE 5-10 % with larger pieces - tools/piece_size_ab.sh).  Small files
            // keep the short pieces (latency to the first contig); FTK_STREAM_PIECE set by hand wins.
            struct stat sb;
            if (!getenv("FTK_STREAM_PIECE") && fstat(fileno(fp), &sb) == 0 && (long long)sb.st_size >= (1ll << 30))
                piece_bytes = 2 * kStreamPiece;
        }
        // a text stream's GPU pieces go up straight from the read buffer too (FTK_TEXT_DIRECT_UP=0: staged in the buffer
        // sets' own page-locked memory by a copy of the producer's - 0.8 ms per 48 MB piece on 16 threads, and in the
        // way of the host threads' inflate jobs)
        static const bool text_direct = !(getenv("FTK_TEXT_DIRECT_UP") && atoi(getenv("FTK_TEXT_DIRECT_UP")) == 0) &&
                                        !(getenv("FTK_DEVICE_INFLATE") && atoi(getenv("FTK_DEVICE_INFLATE")) == 0);
        if (!bam && device >= 0 && text_direct && have_hip_device()) buf.pinned = ahead.pinned = true;
    }
    first_piece_off = ftell(fp);
    const size_t n = fill(buf, 0);
    size_t bsize = 0;
    const bool bgzf = n >= 18 && gzip_header(buf.data(), n, 0, &bsize) && bsize;
    bool ok;
    if (!bgzf) {
        // not block-compressed (plain gzip): no block parallelism to stream; decode whole and hand out per contig
        drain_ahead();
        fclose(fp);
        fp = nullptr;
        ftk_fragtable* whole = nullptr;
        const int rc = bam ? ftk_bam_decode(path.c_str(), has_only ? only.c_str() : nullptr, n_threads, &whole)
                           : ftk_fragfile_decode(path.c_str(), has_only ? only.c_str() : nullptr, n_threads, &whole);
        if (rc != FTK_OK) {
            fail(rc, g_decode_err.c_str());
        } else {
            bed6 = whole->bed6;
            skipped[0] += whole->skipped[0];
            skipped[1] += whole->skipped[1];
            if (bam) {
                std::lock_guard<std::mutex> lk(mu);
                for (auto& ct : whole->contigs) { ref_names.push_back(ct.name); ref_lens.push_back(ct.length); }
                header_ready = true;
                cv.notify_all();
            }
            for (auto& ct : whole->contigs) {
                Contig copy;
                copy.name = ct.name;
                copy.length = ct.length;
                const Packed& p = ct.p;
                copy.c.start.assign(p.start, p.start + p.rows);
                copy.c.end.assign(p.end, p.end + p.rows);
                copy.c.mapq.assign(p.mapq, p.mapq + p.rows);
                copy.c.strand.assign(p.strand, p.strand + p.rows);
                if (p.r1s) { copy.c.r1s.assign(p.r1s, p.r1s + p.rows); copy.c.r1e.assign(p.r1e, p.r1e + p.rows); }
                if (p.ord) copy.c.ord.assign(p.ord, p.ord + p.rows);
                if (p.rows == 0) continue;
                if (!emit(std::move(copy))) break;
            }
            delete whole;
        }
        ok = true;
    } else {
        static const bool dev_bam_parse = !(getenv("FTK_DEVICE_BAM_PARSE") && atoi(getenv("FTK_DEVICE_BAM_PARSE")) == 0) &&
                                          !(getenv("FTK_DEVICE_INFLATE") && atoi(getenv("FTK_DEVICE_INFLATE")) == 0);
        const bool bam_on_device = bam && inflate_device >= 0 && dev_bam_parse;
        ok = bam ? (bam_on_device ? run_bam_device(buf, n) : run_bam(buf, n)) : device >= 0 ? run_text_device(buf, n) : run_text(buf, n);
        bool stopped;
        {
            std::lock_guard<std::mutex> lk(mu);
            stopped = stop || err != FTK_OK;
        }
        if (!ok && want_host_restart && !stopped && bam) {
            // (see run_bam_device) the file once more on the host decoder; contigs handed out already are skipped
            drain_ahead();
            if (pstream) (void)hipStreamSynchronize(pstream);
            want_host_restart = false;
            has_region = false;  // (the host decoder hands out the whole contig)
            read_end = -1;
            partial_tail_ok = false;
            first_skip = 0;
            ahead_ok = !has_only;
            rewind(fp);
            const size_t n2 = fill(buf, 0);
            ok = run_bam(buf, n2);
        } else if (!ok && want_host_restart && !stopped) {
            // (see want_host_restart) the same range of the file once more, inflated by the host threads
            drain_ahead();
            (void)hipStreamSynchronize(pstream);
            host_inflate_only = true;
            has_region = false;  // (the second pass hands out the whole contig)
            read_end = -1;
            partial_tail_ok = false;
            first_skip = 0;
            rewind(fp);
            if (has_only) {
                const IndexSpan sp = index_lookup(index_path_of(path, false), false, only, -1);
                if (sp.usable && sp.present && !seek_to(sp)) { read_end = -1; partial_tail_ok = false; first_skip = 0; rewind(fp); }
            }
            first_piece_off = ftell(fp);
            const size_t n2 = fill(buf, 0);
            ok = run_text_device(buf, n2);
        }
    }
    (void)ok;
    flush();
    std::lock_guard<std::mutex> lk(mu);
    finished = true;
    header_ready = true;
    cv.notify_all();
}

void ftk_fragstream::run() {
    try {
        run_guarded();
        return;
    } catch (const std::exception& ex) {
        fail(FTK_ERR_OOM, ex.what());
    }
    std::lock_guard<std::mutex> lk(mu);
    finished = true;
    header_ready = true;
    cv.notify_all();
}

namespace {
struct StageClock {  // where the producer thread of the streaming decoder spends its time (ftk_fragstream_stage_ms;
                     // FTK_DECODE_TIMING=1 also prints it)
    bool on = getenv("FTK_DECODE_TIMING") != nullptr;
    ftk_fragstream* owner;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    explicit StageClock(ftk_fragstream* s) : owner(s) {}
    void lap(int k) {
        auto now = std::chrono::steady_clock::now();
        acc[k] += std::chrono::duration<double, std::milli>(now - t).count();
        t = now;
    }
    void report(const char* what) {
        {
            std::lock_guard<std::mutex> lk(owner->mu);
            for (int k = 0; k < 6; ++k) owner->stage_ms[k] = acc[k];
        }
        if (!on) return;
        fprintf(stderr, "[ftk stream %s] read %.1f  inflate %.1f  parse %.1f  merge %.1f  emit(pack+queue) %.1f  other %.1f ms\n",
                what, acc[0], acc[1], acc[2], acc[3], acc[4], acc[5]);
    }
};
}  // namespace

bool ftk_fragstream::run_text(RawBuf& buf, size_t n) {
    StageClock clk(this);
    std::vector<Block> blocks;
    RawBuf text;                  // carry (incomplete last line) + this piece's inflated text
    size_t text_carry = 0;
    bool layout_known = false;
    Contig cur;
    bool have_cur = false;
    std::set<std::string> seen;
    bool eof = n < last_want;
    for (;;) {
        size_t used = 0, total = 0;
        if (!whole_blocks(buf.data(), n, eof, &blocks, &used, &total)) return fail(FTK_ERR_FORMAT, "corrupt BGZF block");
        if (!text.reserve(text_carry + total + 1)) return fail(FTK_ERR_OOM, "out of host memory");
        clk.lap(5);
        if (!blocks.empty() && inflate_block_list(buf.data(), blocks, n_threads, text.data() + text_carry) != FTK_OK)
            return fail(FTK_ERR_FORMAT, "BGZF inflate failed");
        clk.lap(1);
        const char* b = (const char*)text.data();
        const char* e = b + text_carry + total;
        if (first_skip) {  // after an index seek: the contig starts inside the first block
            b += std::min<size_t>(first_skip, (size_t)(e - b));
            first_skip = 0;
        }
        if (!layout_known) {  // io/alignment.py:143-156: BED6 when the first data row has > 5 columns
            const char* q = b;
            while (q < e) {
                const char* nl = (const char*)memchr(q, '\n', (size_t)(e - q));
                const char* le = nl ? nl : e;
                if (le > q && *q != '#') {
                    int tabs = 0;
                    for (const char* x = q; x < le; ++x) tabs += (*x == '\t');
                    bed6 = (tabs + 1) > 5;
                    layout_known = true;
                    break;
                }
                if (!nl) break;
                q = nl + 1;
            }
        }
        // complete lines only; the rest waits for the next piece
        const char* last = e;
        if (!eof) {
            while (last > b && last[-1] != '\n') --last;
        }
        std::vector<Run> runs;
        if (last > b) parse_text_parallel(b, last, bed6, has_only ? only.c_str() : nullptr, n_threads, &runs);
        clk.lap(2);
        for (auto& r : runs) {
            if (have_cur && r.name != cur.name) {
                clk.lap(3);
                if (!emit(std::move(cur))) return false;
                clk.lap(4);
                cur = Contig{};
                have_cur = false;
            }
            if (!have_cur) {
                if (!seen.insert(r.name).second)
                    return fail(FTK_ERR_UNSORTED, ("contig " + r.name + " appears in two separate runs: the file is not sorted").c_str());
                cur.name = r.name;
                have_cur = true;
            }
            cur.parts.push_back(std::move(r.c));
        }
        clk.lap(3);
        text_carry = (size_t)(e - last);
        if (text_carry) memmove(text.data(), last, text_carry);
        if (eof) break;
        const size_t raw_carry = n - used;
        if (raw_carry) memmove(buf.data(), buf.data() + used, raw_carry);
        clk.lap(5);
        n = fill(buf, raw_carry);
        clk.lap(0);
        eof = n - raw_carry < last_want;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (stop) return false;
        }
    }
    clk.lap(5);
    if (have_cur && !emit(std::move(cur))) return false;
    clk.lap(4);
    clk.report("text");
    return true;
}

namespace {
// One of the two buffer sets of the device row parser: page-locked host text (the inflate target and the
// DMA source), the device copy, the kernels' scratch and outputs, and the summary that comes back.
struct DevSet {
    uint8_t* h_text = nullptr;  // page-locked twin of d_text: only where the host touches a piece's text (its own share
                                // of the inflate, a piece of odd rows, the host-parse streams) - see ensure_host_text
    size_t h_text_cap = 0;
    uint8_t* d_text = nullptr;
    size_t cap = 0, max_lines = 0;
    void* d_blocks = nullptr;   // the row parser's scan state (ftk::textparse_scratch_bytes)
    int32_t *d_s = nullptr, *d_e = nullptr;
    uint8_t *d_q = nullptr, *d_t = nullptr;
    ftk::TextSummary* d_sum = nullptr;
    ftk::TextSummary* h_sum = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t front = nullptr;  // the piece's bytes are on the device, inflated, CRCs computed (the set's own stream)
    hipEvent_t freed = nullptr;  // the appends that read the set's columns last have run (parse stream)
    bool freed_valid = false;
    bool pending = false;
    bool host_only = false;   // the piece was not sent to the device (4 GB or more: the kernels index with 32 bits)
    size_t off = 0, len = 0;  // the launched range of h_text (complete lines)
    bool cut_tail = false;    // last piece of an index-driven read that may stop inside a row (ignore that one row)
    // pieces inflated on the device (FTK_DEVICE_INFLATE): compressed bytes, block table, per-block CRCs, status
    bool inflated = false;
    uint8_t* d_comp = nullptr;
    uint8_t* h_comp = nullptr;  // page-locked copy of the compressed piece (BAM path; text pieces stage in h_text)
    size_t h_comp_cap = 0;
    size_t comp_cap = 0, tab_cap = 0, n_tab = 0;
    ftk::InflateBlock *d_tab = nullptr, *h_tab = nullptr;
    uint32_t *d_crc = nullptr, *h_crc = nullptr, *want_crc = nullptr;  // want_crc: the blocks' trailers (plain host memory)
    ftk::InflateStatus *d_ist = nullptr, *h_ist = nullptr;

    // BAM pieces parsed on the device (ftk_bamparse.hip): the extra row columns, the stretch scratch, the summary
    int32_t *d_r1s = nullptr, *d_r1e = nullptr, *d_ref = nullptr;
    uint32_t* d_stretch = nullptr;
    size_t stretch_words = 0, bam_rows = 0;
    ftk::BamSummary *d_bsum = nullptr, *h_bsum = nullptr;
    void release_bam() {
        for (void* q : {(void*)d_r1s, (void*)d_r1e, (void*)d_ref, (void*)d_stretch, (void*)d_bsum})
            if (q) (void)hipFree(q);
        if (h_bsum) (void)hipHostFree(h_bsum);
        d_r1s = d_r1e = d_ref = nullptr;
        d_stretch = nullptr;
        d_bsum = h_bsum = nullptr;
        stretch_words = bam_rows = 0;
    }
    // call after ensure(): columns for max_lines rows, stretch scratch for `bytes` of records
    bool ensure_bam(size_t bytes, uint32_t stretch_bytes) {
        const size_t words = ftk::bam_stretch_words(bytes, stretch_bytes);
        if (bam_rows >= max_lines && stretch_words >= words && d_bsum) return true;
        release_bam();
        const bool ok = hipMalloc((void**)&d_r1s, max_lines * 4) == hipSuccess && hipMalloc((void**)&d_r1e, max_lines * 4) == hipSuccess &&
                        hipMalloc((void**)&d_ref, max_lines * 4) == hipSuccess &&
                        hipMalloc((void**)&d_stretch, (words + words / 4) * 4) == hipSuccess &&
                        hipMalloc((void**)&d_bsum, sizeof(ftk::BamSummary)) == hipSuccess &&
                        hipHostMalloc((void**)&h_bsum, sizeof(ftk::BamSummary), hipHostMallocDefault) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            release_bam();
            return false;
        }
        bam_rows = max_lines;
        stretch_words = words + words / 4;
        return true;
    }

    void release_inflate() {
        for (void* q : {(void*)d_comp, (void*)d_tab})  // (d_crc / d_ist lie in d_tab's block, h_crc / h_ist in h_tab's)
            if (q) (void)hipFree(q);
        if (h_tab) (void)hipHostFree(h_tab);
        pinned_unmap(h_comp);
        free(want_crc);
        d_comp = h_comp = nullptr; d_tab = h_tab = nullptr; d_crc = h_crc = want_crc = nullptr; d_ist = h_ist = nullptr;
        comp_cap = tab_cap = h_comp_cap = 0;
    }
    void release() {
        release_inflate();
        release_bam();
        if (h_text) pinned_unmap(h_text);
        h_text_cap = 0;
        if (h_sum) (void)hipHostFree(h_sum);
        for (void* q : {(void*)d_text, (void*)d_blocks, (void*)d_s, (void*)d_e, (void*)d_q, (void*)d_t, (void*)d_sum})
            if (q) (void)hipFree(q);
        for (hipEvent_t ev : {done, front, freed})
            if (ev) (void)hipEventDestroy(ev);
        *this = DevSet{};
    }
    bool ensure_host_comp(size_t comp_bytes) {
        if (comp_bytes <= h_comp_cap) return true;
        if (h_comp) pinned_unmap(h_comp);
        h_comp = nullptr;
        h_comp_cap = comp_bytes + comp_bytes / 4 + 4096;
        if ((h_comp = (uint8_t*)pinned_map(h_comp_cap)) == nullptr) {
            h_comp_cap = 0;
            return false;
        }
        return true;
    }
    // room for a piece of comp_bytes of BGZF data in n_blocks blocks
    bool ensure_inflate(size_t comp_bytes, size_t n_blocks) {
        bool ok = true;
        if (comp_bytes + 64 > comp_cap) {
            if (d_comp) (void)hipFree(d_comp);
            d_comp = nullptr;
            comp_cap = comp_bytes + comp_bytes / 4 + 4096;
            ok = hipMalloc((void**)&d_comp, comp_cap) == hipSuccess;
        }
        if (ok && (n_blocks > tab_cap || !d_ist)) {  // (also a piece without a complete block: the status words are still used)
            const size_t cc = comp_cap, hc = h_comp_cap;
            uint8_t *keep = d_comp, *keep_h = h_comp;
            d_comp = h_comp = nullptr;
            release_inflate();
            d_comp = keep;
            comp_cap = cc;
            h_comp = keep_h;
            h_comp_cap = hc;
            tab_cap = n_blocks + n_blocks / 4 + 64;
            want_crc = (uint32_t*)malloc(tab_cap * 4);
            // block table | CRCs | status, ONE device block and ONE page-locked block (d_tab / h_tab are their bases): a
            // small hipHostMalloc costs 1-8 ms in a process's first pass, a small hipMalloc ~1 ms, and a text stream's
            // twelve sets made three of each
            auto up = [](size_t v) { return (v + 255) / 256 * 256; };
            const size_t o_crc = up(tab_cap * sizeof(ftk::InflateBlock)), o_ist = o_crc + up(tab_cap * 4);
            const size_t all = o_ist + up(sizeof(ftk::InflateStatus));
            ok = want_crc && hipMalloc((void**)&d_tab, all) == hipSuccess &&
                 hipHostMalloc((void**)&h_tab, all, hipHostMallocDefault) == hipSuccess;
            if (ok) {
                d_crc = (uint32_t*)((char*)d_tab + o_crc);
                d_ist = (ftk::InflateStatus*)((char*)d_tab + o_ist);
                h_crc = (uint32_t*)((char*)h_tab + o_crc);
                h_ist = (ftk::InflateStatus*)((char*)h_tab + o_ist);
            }
        }
        if (!ok) {
            (void)hipGetLastError();
            release_inflate();
        }
        return ok;
    }
    // page-locked room for `bytes` of text on the host side (page-locking 250 MB takes ~25 ms: a text stream whose
    // pieces stay on the device never pays it - the first whole-genome pass of a process spent 0.2 s here for its
    // sets)
    bool ensure_host_text(size_t bytes) {
        if (bytes <= h_text_cap) return true;
        if (h_text) pinned_unmap(h_text);
        h_text = nullptr;
        h_text_cap = std::max(bytes + bytes / 4 + 4096, cap);
        if ((h_text = (uint8_t*)pinned_map(h_text_cap)) == nullptr) {
            h_text_cap = 0;
            return false;
        }
        return true;
    }
    // room for `bytes` of text; false: out of (page-locked or device) memory
    bool ensure(size_t bytes, bool with_host_text = true) {
        if (bytes <= cap) return !with_host_text || ensure_host_text(bytes);
        release();
        const size_t want = bytes + bytes / 4 + 4096;
        const size_t lines = want / 10 + 1;  // a plain row is at least 10 bytes; more lines -> the host parses the piece
        bool ok = (!with_host_text || ensure_host_text(want)) &&
                  hipHostMalloc((void**)&h_sum, sizeof(ftk::TextSummary), hipHostMallocDefault) == hipSuccess &&
                  hipMalloc((void**)&d_text, want) == hipSuccess &&
                  hipMalloc((void**)&d_blocks, ftk::textparse_scratch_bytes(want)) == hipSuccess &&
                  hipMalloc((void**)&d_s, lines * 4) == hipSuccess && hipMalloc((void**)&d_e, lines * 4) == hipSuccess &&
                  hipMalloc((void**)&d_q, lines) == hipSuccess && hipMalloc((void**)&d_t, lines) == hipSuccess &&
                  hipMalloc((void**)&d_sum, sizeof(ftk::TextSummary)) == hipSuccess &&
                  hipEventCreateWithFlags(&done, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&front, hipEventDisableTiming) == hipSuccess &&
                  hipEventCreateWithFlags(&freed, hipEventDisableTiming) == hipSuccess;
        if (!ok) {
            (void)hipGetLastError();
            release();
            return false;
        }
        cap = want;
        max_lines = lines;
        return true;
    }
};
// The two buffer sets of a finished stream wait here for the next one: allocating them costs ~100 ms (400 MB
// of page-locked memory, ~1 GB of device memory, the frees synchronise the device) - more than a small file
// takes to decode.  At most twelve idle sets are kept (per process, any device): the ring of a text stream.
struct DevSetPool {
    std::mutex mu;
    std::vector<std::pair<int, DevSet>> idle;
    DevSet take(int device) {
        std::lock_guard<std::mutex> lk(mu);
        for (size_t i = 0; i < idle.size(); ++i)
            if (idle[i].first == device) {
                DevSet s = idle[i].second;
                idle.erase(idle.begin() + i);
                return s;
            }
        return DevSet{};
    }
    void give(int device, DevSet& s) {
        s.pending = false;
        s.freed_valid = false;  // (the giver has synchronised its streams)
        {
            std::lock_guard<std::mutex> lk(mu);
            if (s.cap && idle.size() < 12) {
                idle.emplace_back(device, s);
                s = DevSet{};
                return;
            }
        }
        s.release();
    }
    size_t trim() {  // release every idle set; returns their page-locked + device bytes (text buffers only: a lower bound)
        std::vector<std::pair<int, DevSet>> drop;
        {
            std::lock_guard<std::mutex> lk(mu);
            drop.swap(idle);
        }
        size_t n = 0;
        for (auto& d : drop) {
            (void)hipSetDevice(d.first);
            n += d.second.cap + d.second.h_text_cap + d.second.h_comp_cap + d.second.comp_cap;
            d.second.release();
        }
        return n;
    }
};
DevSetPool& devset_pool() {
    static DevSetPool* p = new DevSetPool();  // leaked: the driver frees at process exit
    return *p;
}

// HIP streams of finished decoder streams wait here for the next one: creating one costs ~1 ms, destroying it as
// much, a decoder stream uses five - a fifth of the time a small file takes from disk to results.  A stream is idle
// (synchronised) when it is given back.
struct StreamPool {
    std::mutex mu;
    std::vector<std::pair<int, hipStream_t>> idle;
    hipStream_t take(int device) {  // nullptr: cannot create one
        {
            std::lock_guard<std::mutex> lk(mu);
            for (size_t i = 0; i < idle.size(); ++i)
                if (idle[i].first == device) {
                    hipStream_t s = idle[i].second;
                    idle.erase(idle.begin() + i);
                    return s;
                }
        }
        hipStream_t s = nullptr;
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            (void)hipGetLastError();
            return nullptr;
        }
        return s;
    }
    void fill_to(int device, int n) {  // idle streams of `device` up to n (a helper thread's job: see FrontStreams::prefill)
        for (;;) {
            {
                std::lock_guard<std::mutex> lk(mu);
                int have = 0;
                for (auto& e : idle) have += e.first == device;
                if (have >= n || idle.size() >= 16) return;
            }
            hipStream_t s = nullptr;
            if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
                (void)hipGetLastError();
                return;
            }
            give(device, s);
        }
    }
    void give(int device, hipStream_t s) {
        if (!s) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (idle.size() < 16) {
                idle.emplace_back(device, s);
                return;
            }
        }
        (void)hipStreamDestroy(s);
    }
    size_t trim() {
        std::vector<std::pair<int, hipStream_t>> drop;
        {
            std::lock_guard<std::mutex> lk(mu);
            drop.swap(idle);
        }
        for (auto& d : drop) {
            (void)hipSetDevice(d.first);
            (void)hipStreamDestroy(d.second);
        }
        return drop.size();
    }
};
StreamPool& stream_pool() {
    static StreamPool* p = new StreamPool();  // leaked: the driver frees at process exit
    return *p;
}

// The front streams of a decoder stream's ring, created when a piece first needs one.  With sixteen hardware queues
// (GPU_MAX_HW_QUEUES, see _hardware_queues) a NEW stream costs 5.5 ms - its queue is set up with it - and a decoder
// stream that opened its whole ring up front spent 72 of the 80 ms a small file takes in a fresh process on thirteen
// hipStreamCreateWithFlags (profiles/r4_cold_start.txt); a file of one piece needs one.  Streams of earlier decoder
// streams come back from the pool at no cost, so a warm process sees no difference.  If a stream cannot be created
// the piece runs on `fallback` (the parse stream): ordering is by events, so that only serialises it.
template <int N>
struct FrontStreams {
    int device = -1;
    hipStream_t fallback = nullptr;
    hipStream_t s[N] = {};
    hipStream_t get(int k) {
        if (!s[k] && (s[k] = stream_pool().take(device)) == nullptr) return fallback;
        return s[k];
    }
    // A file of many pieces will use the whole ring: a helper thread creates the streams the pool lacks while the
    // producer reads and launches the first pieces (each get() then finds one idle instead of spending 5.5 ms).
    std::thread filler;
    void prefill(int n) {
        if (n <= 0 || filler.joinable()) return;
        const int dev = device;
        filler = std::thread([dev, n] {
            if (hipSetDevice(dev) == hipSuccess) stream_pool().fill_to(dev, std::min(n, N));
        });
    }
    void settle_and_give() {  // (the caller has set the device)
        if (filler.joinable()) filler.join();
        for (auto& q : s)
            if (q) {
                (void)hipStreamSynchronize(q);
                stream_pool().give(device, q);
                q = nullptr;
            }
    }
};
}  // namespace

bool ftk_fragstream::emit_device(Contig&& ct) {
    DevColumns& d = *ct.dev;
    if (hipEventCreateWithFlags(&d.ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(d.ready, pstream) != hipSuccess) {
        (void)hipGetLastError();
        return fail(FTK_ERR_HIP, "cannot record the contig's ready event");
    }
    std::unique_ptr<ftk_fragtable> t(new ftk_fragtable());
    t->bed6 = bed6;
    ct.p.rows = d.rows;
    ct.p.start = d.start;
    ct.p.end = d.end;
    ct.p.mapq = d.mapq;
    ct.p.strand = d.strand;
    emitted_names.insert(ct.name);
    t->contigs.push_back(std::move(ct));
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return stop || ready.size() < max_queued; });
    if (stop) return false;
    ready.push_back(t.release());
    cv.notify_all();
    return true;
}

// run_text with the row parser on the GPU.  Per piece: inflate into page-locked text (host threads), one
// DMA, four kernels, 8 KB of summary back - all asynchronous on the parse stream, and while they run the
// host already inflates the next piece into the other buffer set.  When a piece's summary says "plain
// rows only" its columns are appended to the current contig device-to-device, split at the contig runs the
// kernel listed (the names are read from the host copy of the text); any other piece goes through the
// host's field-rule parser (parse_text_parallel) and its columns are uploaded - same rows either way.
bool ftk_fragstream::run_text_device(RawBuf& buf, size_t n) {
    StageClock clk(this);
    if (hipSetDevice(device) != hipSuccess || (!pstream && (pstream = stream_pool().take(device)) == nullptr)) {
        (void)hipGetLastError();
        return fail(FTK_ERR_HIP, "cannot create the parse stream");
    }
    // Four buffer sets in a ring, settled two pieces behind the one being launched: while the parse stream works on
    // piece k-1 (set-up, row parser, the appends of k-2), the FRONT of piece k - compressed bytes up, inflate and CRC
    // kernels, nothing that depends on another piece - runs on the set's own stream beside piece k-1's front.
    // With the host threads taking a share of the pieces (below) the ring is longer: a host piece's parse - and, the
    // parse stream being in file order, those of the pieces behind it - goes onto the stream when its inflate is done,
    // at the latest kHostLag pieces later.
    constexpr int kSets = 12, kHostLag = 8;
    DevSet sets[kSets];
    for (auto& S : sets) S = devset_pool().take(device);
    FrontStreams<kSets> fstream;
    fstream.device = device;
    fstream.fallback = pstream;
    fstream.prefill(std::min(pieces_expected(), kSets) - 1);
    // FTK_DECODE_TIMING: the device time of every piece's front (copy up + inflate + CRC) and back (set-up + rows)
    hipEvent_t tev[kSets][5] = {};  // front start, front end, back start, back end, bytes up
    double front_ms = 0, front_max = 0, back_ms = 0, back_max = 0;
    size_t n_timed = 0;
    const auto t_begin = std::chrono::steady_clock::now();
    auto now_ms = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    std::string trail;  // FTK_DECODE_TIMING=2: one line per settled piece, and the producer's own steps
    const bool trail_on = clk.on && atoi(getenv("FTK_DECODE_TIMING")) >= 2;
    auto mark = [&](int piece, const char* what) {
        if (!trail_on) return;
        char line[120];
        snprintf(line, sizeof line, "  piece %d %s at %.2f ms\n", piece, what, now_ms());
        trail += line;
    };
    if (clk.on)
        for (auto& row : tev)
            for (auto& ev : row) (void)hipEventCreate(&ev);
    struct Cleanup {
        DevSet* s;
        int device;
        hipStream_t stream;
        FrontStreams<kSets>* fs;
        hipEvent_t (*tev)[5];
        ~Cleanup() {
            fs->settle_and_give();
            (void)hipStreamSynchronize(stream);  // nothing in flight touches the sets any more
            for (int k = 0; k < kSets; ++k) {
                devset_pool().give(device, s[k]);
                for (hipEvent_t ev : tev[k])
                    if (ev) (void)hipEventDestroy(ev);
            }
        }
    } cleanup{sets, device, pstream, &fstream, tev};
    std::vector<Block> blocks;
    size_t carry = 0;
    const uint8_t* carry_src = nullptr;
    bool layout_known = false;
    Contig cur;
    bool have_cur = false;
    bool saw_other = false;
    std::set<std::string> seen;
    size_t gpu_pieces = 0, host_pieces = 0;
    // The BGZF blocks are inflated on the GPU too (ftk_inflate.hip): the host only reads the file and copies it into
    // page-locked memory.  FTK_DEVICE_INFLATE=0 keeps the inflate on the host threads (libdeflate / zlib).
    static const bool dev_inflate_env = !(getenv("FTK_DEVICE_INFLATE") && atoi(getenv("FTK_DEVICE_INFLATE")) == 0);
    const bool dev_inflate = dev_inflate_env && !host_inflate_only;
    // The host threads CAN inflate pieces beside the GPU (a host piece's text goes up in one DMA before its parse; they
    // take the next piece whenever their previous one is done, so the split follows the two rates).  That was worth 18 %
    // when the chip turned fragment rows over at ~30 GB/s (round 3) and 3-5 % at 87 GB/s (round 4); with round 5's
    // symbol loop (185 GB/s) the pieces they take arrive LATER than the GPU would have had them:
    // whole-genome pass 0.089-0.109 s with, 0.078-0.089 s without (tools/env_ab.sh text).  Off by default since.
    // FTK_TEXT_HOST_SHARE=<n>: 0 none (the default), 1 as the threads are free, n > 1 every n-th piece.
    static const int host_share_env = [] {
        const char* e = getenv("FTK_TEXT_HOST_SHARE");
        return e ? atoi(e) : -1;
    }();
    const int host_share = !dev_inflate ? 0 : host_share_env >= 0 ? host_share_env : 0;
    static const int lag_env = [] {
        const char* e = getenv("FTK_TEXT_LAG");
        return e ? atoi(e) : -1;
    }();
    // (pieces the producer lets the GPU fall behind before it waits: FTK_TEXT_LAG; without host pieces 2 until round 5 -
    // 0.078-0.089 s per genome pass - now 4: 0.074-0.085 s; 6 and 9 the same)
    const int kLag = lag_env >= 1 ? std::min(lag_env, kHostLag + 1) : host_share > 0 ? kHostLag + 1 : 4;
    struct PieceMeta {
        bool on_host = false, eof = false, has_prev = false, back_done = true;
        size_t total = 0;
        uint32_t first_skip = 0;
    } meta[kSets];
    std::future<int> host_job[kSets];
    struct JobGuard {  // no job outlives the buffers it works on
        std::future<int>* j;
        ~JobGuard() {
            for (int i = 0; i < kSets; ++i)
                if (j[i].valid()) (void)j[i].get();
        }
    } job_guard{host_job};
    double t_jobwait = 0;
    size_t host_inflated = 0;
    // (a GPU piece's compressed bytes go up straight from the page-locked read buffer; fill() lets that buffer rest
    // until the copy is done - see buf_in_flight)
    struct RestGuard {
        ftk_fragstream* s;
        ~RestGuard() { s->drop_resting(); }
    } rest_guard{this};  // (waits for the copies that still read a resting buffer)
    long long piece_off = first_piece_off;  // where buf's first byte lies in the file (-1: unknown)
    int settled = 0, backs = 0;  // pieces settled / pieces whose back is on the parse stream
    int last_host_set = -1;  // the set of the piece the host threads took last
    auto host_takes = [&](int k) -> bool {
        if (host_share <= 0 || k == 0) return false;
        if (host_share > 1) return (k % host_share) == host_share - 1;
        // (only with a backlog on the GPU: the first pieces of a stream - all of a small file - are back sooner from the
        // chip, 3 ms a piece against the threads' 12)
        if (k - settled < 3) return false;
        // (and only pieces of three of the twelve sets: a host piece needs 250 MB of page-locked text behind its set -
        // 25 ms to lock - and at the two rates the threads take about every fifth piece anyway)
        if ((k & 3) != 0) return false;
        return last_host_set < 0 || !host_job[last_host_set].valid() ||
               host_job[last_host_set].wait_for(std::chrono::seconds(0)) == std::future_status::ready;
    };

    // one contig run of a piece: n rows at the given column pointers (device or host)
    auto take_run = [&](const std::string& name, const int32_t* s0, const int32_t* e0, const uint8_t* q0, const uint8_t* t0,
                        size_t rows, hipMemcpyKind kind) -> bool {
        if (has_only && name != only) {
            saw_other = true;  // (rows behind the wanted contig's: a region read is complete)
            return true;
        }
        if (host_inflate_only && emitted_names.count(name)) return true;  // second pass: handed out by the first
        if (have_cur && name != cur.name) {
            if (!emit_device(std::move(cur))) return false;
            cur = Contig{};
            have_cur = false;
        }
        if (!have_cur) {
            if (!seen.insert(name).second)
                return fail(FTK_ERR_UNSORTED, ("contig " + name + " appears in two separate runs: the file is not sorted").c_str());
            cur.name = name;
            cur.dev.reset(new DevColumns());
            cur.dev->device = device;
            have_cur = true;
        }
        if (!cur.dev->append(s0, e0, q0, t0, rows, kind, pstream)) return fail(FTK_ERR_OOM, "out of device memory for the contig's columns");
        return true;
    };

    auto collect_rows = [&](DevSet& S) -> bool {
        if (!S.host_only && hipEventSynchronize(S.done) != hipSuccess) {
            (void)hipGetLastError();
            return fail(FTK_ERR_HIP, "the device row parser failed");
        }
        S.pending = false;
        const ftk::TextSummary& sum = *S.h_sum;
        if (S.inflated) {
            if (S.h_ist->n_bad)
                return fail(FTK_ERR_FORMAT, ("BGZF inflate failed: a block did not decode to its ISIZE (device inflate, reason " +
                                             std::to_string(S.h_ist->reason) + ")").c_str());
            for (size_t i = 0; i < S.n_tab; ++i)
                if (S.h_crc[i] != S.want_crc[i]) return fail(FTK_ERR_FORMAT, "BGZF block CRC mismatch (device inflate)");
            if (sum.carry_overflow) {  // a line too long for the device carry: the host-inflate pass takes the file
                want_host_restart = true;
                return false;
            }
            if (sum.text_len == 0) return true;
            const bool drop_last = S.cut_tail && sum.last_line_bad && sum.n_bad == 1;  // the row the read stopped in
            const size_t n_rows = (size_t)sum.n_lines - (drop_last ? 1 : 0);
            const bool plain_d = !sum.overflow && (sum.n_bad == 0 || drop_last) && sum.n_runs >= 1 &&
                                 sum.n_runs <= (unsigned)ftk::kTextNamedRuns && !sum.name_overflow && sum.n_lines <= S.max_lines;
            if (plain_d) {
                ++gpu_pieces;
                std::vector<std::pair<unsigned, unsigned>> runs(sum.n_runs);  // (first line, slot)
                for (unsigned r = 0; r < sum.n_runs; ++r) runs[r] = {sum.run_line[r], r};
                std::sort(runs.begin(), runs.end());
                for (size_t r = 0; r < runs.size(); ++r) {
                    const size_t l0 = runs[r].first, l1 = r + 1 < runs.size() ? runs[r + 1].first : n_rows;
                    if (!take_run(std::string((const char*)sum.run_name[runs[r].second]), S.d_s + l0, S.d_e + l0, S.d_q + l0,
                                  S.d_t + l0, l1 - l0, hipMemcpyDeviceToDevice))
                        return false;
                }
                return true;
            }
            // anything but plain rows: the host's field-rule parser reads the text (copied back for this piece only)
            ++host_pieces;
            if (!S.ensure_host_text(sum.text_len + 64)) return fail(FTK_ERR_OOM, "out of page-locked memory for a text piece");
            if (hipMemcpyAsync(S.h_text, S.d_text + sum.text_off, sum.text_len, hipMemcpyDeviceToHost, pstream) != hipSuccess ||
                hipStreamSynchronize(pstream) != hipSuccess) {
                (void)hipGetLastError();
                return fail(FTK_ERR_HIP, "cannot copy a text piece back");
            }
            std::vector<Run> runs;
            const char* tb = (const char*)S.h_text;
            parse_text_parallel(tb, tb + sum.text_len, bed6, has_only ? only.c_str() : nullptr, n_threads, &runs);
            for (auto& r : runs)
                if (!take_run(r.name, r.c.start.data(), r.c.end.data(), r.c.mapq.data(), r.c.strand.data(), r.c.start.size(),
                              hipMemcpyHostToDevice))
                    return false;
            return true;
        }
        const char* b = (const char*)S.h_text + S.off;
        const bool drop_last_h = S.cut_tail && sum.last_line_bad && sum.n_bad == 1;  // the row an index-driven read stopped in
        const bool plain = !S.host_only && !sum.overflow && (sum.n_bad == 0 || drop_last_h) && sum.n_runs >= 1 &&
                           sum.n_runs <= (unsigned)ftk::kTextMaxRuns && sum.n_lines <= S.max_lines;
        if (plain) {
            ++gpu_pieces;
            std::vector<std::pair<unsigned, unsigned>> runs(sum.n_runs);
            for (unsigned r = 0; r < sum.n_runs; ++r) runs[r] = {sum.run_line[r], sum.run_off[r]};
            std::sort(runs.begin(), runs.end());
            for (size_t r = 0; r < runs.size(); ++r) {
                const char* nb = b + runs[r].second;
                const char* tab = (const char*)memchr(nb, '\t', S.len - runs[r].second);
                if (!tab) return fail(FTK_ERR_FORMAT, "device row parser: run without a name");
                const size_t l0 = runs[r].first,
                             l1 = r + 1 < runs.size() ? runs[r + 1].first : (size_t)sum.n_lines - (drop_last_h ? 1 : 0);
                if (!take_run(std::string(nb, (size_t)(tab - nb)), S.d_s + l0, S.d_e + l0, S.d_q + l0, S.d_t + l0, l1 - l0,
                              hipMemcpyDeviceToDevice))
                    return false;
            }
        } else {
            ++host_pieces;
            std::vector<Run> runs;
            parse_text_parallel(b, b + S.len, bed6, has_only ? only.c_str() : nullptr, n_threads, &runs);
            for (auto& r : runs)
                if (!take_run(r.name, r.c.start.data(), r.c.end.data(), r.c.mapq.data(), r.c.strand.data(), r.c.start.size(),
                              hipMemcpyHostToDevice))
                    return false;
        }
        return true;
    };

    // settle a piece: its rows to the current contig (device to device, on the parse stream), and behind them the
    // event the set's next front waits for before it overwrites the buffers
    auto collect = [&](DevSet& S) -> bool {
        if (!collect_rows(S)) return false;
        if (clk.on && S.inflated) {
            const int j = (int)(&S - sets);
            float f = 0, bk = 0, gap = 0;
            if (hipEventElapsedTime(&f, tev[j][0], tev[j][1]) == hipSuccess && hipEventElapsedTime(&bk, tev[j][2], tev[j][3]) == hipSuccess) {
                float up = 0;
                (void)hipEventElapsedTime(&gap, tev[j][1], tev[j][2]);
                (void)hipEventElapsedTime(&up, tev[j][0], tev[j][4]);
                char line[200];
                snprintf(line, sizeof line, "  settled at %.1f ms: front %.2f (bytes up %.2f), front end -> back start %.2f, back %.2f ms\n", now_ms(), f, up, gap, bk);
                trail += line;
                front_ms += f; front_max = std::max(front_max, (double)f);
                back_ms += bk; back_max = std::max(back_max, (double)bk);
                ++n_timed;
            } else {
                (void)hipGetLastError();
            }
        }
        S.freed_valid = hipEventRecord(S.freed, pstream) == hipSuccess;
        if (!S.freed_valid) {
            (void)hipGetLastError();
            return fail(FTK_ERR_HIP, "cannot record a buffer set's release");
        }
        return true;
    };
    // Settle, in file order, every piece up to k whose results are already back, and wait for those that are kLag
    // or more behind (their sets come up for reuse).
    auto settle = [&](int k, bool all) -> bool {
        while (settled <= k && settled < backs) {
            DevSet& Q = sets[settled % kSets];
            if (Q.pending) {
                const bool must = all || settled <= k - kLag;
                if (!must) {  // ahead of need only while the consumer is blocked waiting for a contig
                    std::lock_guard<std::mutex> lk(mu);
                    if (!consumer_waiting) break;
                }
                if (!must && !Q.host_only && hipEventQuery(Q.done) != hipSuccess) {
                    (void)hipGetLastError();  // (not ready is no error)
                    break;
                }
                if (!collect(Q)) return false;
            }
            ++settled;
        }
        return true;
    };
    // The back of piece j (the parse stream, piece after piece): a host piece's text up first (on the set's own stream,
    // behind the appends that read the set last), then carry from the previous piece, line ends, rows.
    auto submit_back = [&](int j) -> bool {
        const int sj = j % kSets;
        DevSet& S = sets[sj];
        PieceMeta& M = meta[sj];
        DevSet* P = M.has_prev ? &sets[(j - 1) % kSets] : nullptr;
        bool ok = true;
        if (M.on_host) {
            hipStream_t front = fstream.get(sj);
            const auto t0 = std::chrono::steady_clock::now();
            const int jrc = host_job[sj].valid() ? host_job[sj].get() : (int)FTK_OK;
            t_jobwait += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (jrc != FTK_OK) return fail(FTK_ERR_FORMAT, "BGZF inflate failed or block CRC mismatch (host share of a device stream)");
            ok = (!S.freed_valid || hipStreamWaitEvent(front, S.freed, 0) == hipSuccess) &&
                 (!clk.on || hipEventRecord(tev[sj][0], front) == hipSuccess) &&
                 hipMemsetAsync(S.d_ist, 0, sizeof(ftk::InflateStatus), front) == hipSuccess &&
                 (M.total == 0 || hipMemcpyAsync(S.d_text + ftk::kTextCarryMax, S.h_text + ftk::kTextCarryMax, M.total,
                                                 hipMemcpyHostToDevice, front) == hipSuccess) &&
                 (!clk.on || (hipEventRecord(tev[sj][4], front) == hipSuccess && hipEventRecord(tev[sj][1], front) == hipSuccess)) &&
                 hipEventRecord(S.front, front) == hipSuccess;
        }
        ok = ok && hipMemsetAsync(S.d_sum, 0, sizeof(ftk::TextSummary), pstream) == hipSuccess &&
             hipStreamWaitEvent(pstream, S.front, 0) == hipSuccess && (!clk.on || hipEventRecord(tev[sj][2], pstream) == hipSuccess);
        if (ok) {
            ftk::textparse_launch_inflated(pstream, S.d_text, ftk::kTextCarryMax, (uint32_t)M.total, P ? P->d_text : nullptr,
                                           P ? P->d_sum : nullptr, M.first_skip, M.eof, bed6, S.d_blocks, S.max_lines,
                                           S.d_s, S.d_e, S.d_q, S.d_t, S.d_sum);
            ok = hipGetLastError() == hipSuccess &&
                 hipMemcpyAsync(S.h_sum, S.d_sum, sizeof(ftk::TextSummary), hipMemcpyDeviceToHost, pstream) == hipSuccess &&
                 hipMemcpyAsync(S.h_ist, S.d_ist, sizeof(ftk::InflateStatus), hipMemcpyDeviceToHost, pstream) == hipSuccess &&
                 (S.n_tab == 0 || hipMemcpyAsync(S.h_crc, S.d_crc, S.n_tab * 4, hipMemcpyDeviceToHost, pstream) == hipSuccess) &&
                 (!clk.on || hipEventRecord(tev[sj][3], pstream) == hipSuccess) && hipEventRecord(S.done, pstream) == hipSuccess;
        }
        if (ok && P) {
            // the previous piece's text has been read (the carry): its set may take a new piece behind THIS point,
            // wherever the piece itself was settled
            P->freed_valid = hipEventRecord(P->freed, pstream) == hipSuccess;
            ok = P->freed_valid;
        }
        if (!ok) {
            (void)hipGetLastError();
            return fail(FTK_ERR_HIP, "cannot launch the device row parser");
        }
        S.pending = true;
        M.back_done = true;
        return true;
    };
    // backs up to piece k, in file order; a host piece whose job is still running holds them up until it is kHostLag
    // pieces behind (or `all`)
    auto submit_backs = [&](int k, bool all) -> bool {
        while (backs <= k) {
            const int sj = backs % kSets;
            if (!all && meta[sj].on_host && backs > k - kHostLag && host_job[sj].valid() &&
                host_job[sj].wait_for(std::chrono::seconds(0)) != std::future_status::ready)
                break;
            if (!meta[sj].back_done && !submit_back(backs)) return false;
            ++backs;
        }
        return true;
    };
    bool eof = n < last_want;
    // (region reads) this short read stopped at the linear index's hint, not at the end of the contig's rows: the piece
    // is parsed as one with more behind it, and the rows then say whether to read on
    auto at_soft_end = [&](size_t n_now) {
        return eof && has_region && read_end >= 0 && read_end < hard_read_end && piece_off >= 0 &&
               piece_off + (long long)n_now >= read_end;
    };
    bool soft = at_soft_end(n);
    int k = 0;
    for (;; ++k) {
        size_t used = 0, total = 0;
        if (!whole_blocks(buf.data(), n, eof, &blocks, &used, &total)) return fail(FTK_ERR_FORMAT, "corrupt BGZF block");
        DevSet& S = sets[k % kSets];
        if (S.pending) return fail(FTK_ERR_HIP, "buffer ring out of step");
        if (dev_inflate && total + ftk::kTextCarryMax + 64 < (size_t(1) << 32)) {
            // ---- the piece is inflated ON THE DEVICE: compressed bytes up, one wave per BGZF block, then the carry /
            // line-end set-up and the row parser on the text where it lies; the host never sees the text
            if (!layout_known) {  // BED6 or not: the first data row, from the first blocks inflated here
                std::vector<uint8_t> head;
                size_t nb = 0, bytes = 0;
                while (nb < blocks.size() && bytes < (size_t(1) << 18)) bytes += blocks[nb++].out_len;
                head.resize(bytes + 1);
                std::vector<Block> first(blocks.begin(), blocks.begin() + nb);
                if (nb && inflate_block_list(buf.data(), first, n_threads, head.data()) != FTK_OK)
                    return fail(FTK_ERR_FORMAT, "BGZF inflate failed");
                const char* q = (const char*)head.data() + std::min(first_skip, bytes);
                const char* e = (const char*)head.data() + bytes;
                while (q < e) {
                    const char* nl = (const char*)memchr(q, '\n', (size_t)(e - q));
                    const char* le = nl ? nl : e;
                    if (le > q && *q != '#' && (nl || eof)) {
                        int tabs = 0;
                        for (const char* x = q; x < le; ++x) tabs += (*x == '\t');
                        bed6 = (tabs + 1) > 5;
                        layout_known = true;
                        break;
                    }
                    if (!nl) break;
                    q = nl + 1;
                }
            }
            // h_text also stages the piece's COMPRESSED bytes on their way up: many tiny blocks (or small ISIZE
            // trailers) make `used` larger than the text, so the buffer is sized for whichever is larger
            const bool to_host = host_takes(k);
            // a GPU piece goes up straight from the read buffer when that is page-locked; the host threads work on their
            // own copy of theirs, which they read from the file (page cache) themselves when its offset is known
            hipEvent_t up_ev = (buf.pinned && !to_host) ? take_up_event() : nullptr;
            const bool direct = up_ev != nullptr;
            const bool job_reads = to_host && piece_off >= 0;
            if (!S.ensure(std::max(ftk::kTextCarryMax + total + 64, used + 64), false) || !S.ensure_inflate(used, blocks.size()) ||
                (to_host && (!S.ensure_host_comp(used + 64) || !S.ensure_host_text(ftk::kTextCarryMax + total + 64))) ||
                (!to_host && !direct && !S.ensure_host_text(used + 64)))
                return fail(FTK_ERR_OOM, "out of page-locked / device memory for the text piece");
            clk.lap(5);
            mark(k, to_host ? "(host) blocks listed, buffers ready" : "blocks listed, buffers ready");
            if (!direct && !job_reads) {
                const int nt = std::max(1, std::min(n_threads, (int)(used >> 20) + 1));
                const uint8_t* src = buf.data();
                uint8_t* dst = to_host ? S.h_comp : S.h_text;
                parallel_run(nt, [&](int t) {
                    const size_t a = used * (size_t)t / nt, b2 = used * (size_t)(t + 1) / nt;
                    memcpy(dst + a, src + a, b2 - a);
                });
            }
            for (size_t i = 0; i < blocks.size(); ++i) {
                const Block& bl = blocks[i];
                S.h_tab[i] = {(uint32_t)bl.in_off, (uint32_t)bl.in_len, (uint32_t)(ftk::kTextCarryMax + bl.out_off), (uint32_t)bl.out_len};
                const uint8_t* tr = buf.data() + bl.in_off + bl.in_len;
                S.want_crc[i] = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
            }
            S.n_tab = blocks.size();
            S.cut_tail = eof && !soft && partial_tail_ok;
            S.inflated = true;
            S.host_only = false;
            mark(k, "staged");
            PieceMeta& M = meta[k % kSets];
            M = PieceMeta{};
            M.on_host = to_host;
            M.eof = eof && !soft;
            M.has_prev = k > 0 && sets[(k - 1) % kSets].inflated;
            M.back_done = false;
            M.total = total;
            M.first_skip = (uint32_t)std::min<size_t>(first_skip, total);
            hipStream_t front = fstream.get(k % kSets);
            bool ok = true;
            if (M.on_host) {
                // (the job owns its block list and works from its own copy of the bytes - the staging above - straight
                // into the set's page-locked text buffer; CRCs checked like the GPU's pieces)
                S.n_tab = 0;
                ++host_inflated;
                last_host_set = k % kSets;
                host_job[k % kSets] = std::async(std::launch::async, [bl = blocks, comp = S.h_comp, out = S.h_text + ftk::kTextCarryMax,
                                                                      nt = std::max(1, n_threads - 2), fd = fileno(fp), used,
                                                                      off = piece_off, job_reads] {
                    if (job_reads) {  // its own copy of the compressed bytes, four pread threads
                        std::atomic<int> bad{0};
                        std::vector<std::thread> th;
                        auto part = [&](int t) {
                            size_t a = used * (size_t)t / 4;
                            const size_t e = used * (size_t)(t + 1) / 4;
                            while (a < e) {
                                const ssize_t r = pread(fd, comp + a, e - a, (off_t)(off + (long long)a));
                                if (r <= 0) { bad.store(1); return; }
                                a += (size_t)r;
                            }
                        };
                        for (int t = 1; t < 4; ++t) th.emplace_back(part, t);
                        part(0);
                        for (auto& t : th) t.join();
                        if (bad.load()) return (int)FTK_ERR_IO;
                    }
                    return bl.empty() ? (int)FTK_OK : inflate_block_list(comp, bl, nt, out, true, true);
                });
            } else {
                // front (this set's stream): behind the appends that read the set last, bytes up, inflate, CRC
                // (the compressed bytes go up at once - nothing of the set's previous piece uses d_comp any more - and
                // only the inflate, which overwrites the text the appends may still read, waits for the set's release)
                ok = (!clk.on || hipEventRecord(tev[k % kSets][0], front) == hipSuccess) &&
                     (used == 0 || hipMemcpyAsync(S.d_comp, direct ? buf.data() : S.h_text, used, hipMemcpyHostToDevice, front) == hipSuccess) &&
                     (!direct || hipEventRecord(up_ev, front) == hipSuccess) &&
                     (!S.freed_valid || hipStreamWaitEvent(front, S.freed, 0) == hipSuccess) &&
                     hipMemsetAsync(S.d_ist, 0, sizeof(ftk::InflateStatus), front) == hipSuccess &&
                     (blocks.empty() || hipMemcpyAsync(S.d_tab, S.h_tab, blocks.size() * sizeof(ftk::InflateBlock),
                                                       hipMemcpyHostToDevice, front) == hipSuccess);
                if (direct) buf_in_flight = up_ev;  // (fill() parks the buffer behind it; an unrecorded event reads as done)
                if (ok && clk.on) ok = hipEventRecord(tev[k % kSets][4], front) == hipSuccess;
                if (ok) {
                    ftk::inflate_launch(front, S.d_comp, S.d_tab, (int)blocks.size(), S.d_text, S.d_ist, S.d_crc);
                    ok = hipGetLastError() == hipSuccess && (!clk.on || hipEventRecord(tev[k % kSets][1], front) == hipSuccess) &&
                         hipEventRecord(S.front, front) == hipSuccess;
                }
            }
            if (!ok) {
                const hipError_t he = hipGetLastError();
                return fail(FTK_ERR_HIP, (std::string("cannot launch the device inflate (") + hipGetErrorName(he) + ", piece " +
                                          std::to_string(k) + ", " + std::to_string(blocks.size()) + " blocks, " +
                                          std::to_string(used) + " bytes)").c_str());
            }
            // the backs, in file order: every piece whose text is (about to be) on the device - a host piece when its
            // job is done, or kHostLag pieces later at the latest
            mark(k, "front enqueued");
            if (!submit_backs(k, false)) return false;
            first_skip = 0;
            clk.lap(1);
            mark(k, "backs enqueued");
            // this piece is on its way: settle what is back already, and the piece two back in any case (its set is
            // the one piece k+2 stages into)
            if (!settle(k, false)) return false;
            clk.lap(3);
            mark(k, "settle done");
            if (eof) {
                if (!soft) break;
                // the region's rows so far: complete when a row starts at or behind the region's end (or another
                // contig's rows came); else a row longer than an index window hid the true end - read on
                if (!submit_backs(k, true) || !settle(k, true)) return false;
                bool complete = saw_other;
                if (!complete && have_cur && cur.dev->rows) {
                    int32_t last = 0;
                    if (hipMemcpyAsync(&last, cur.dev->start + (cur.dev->rows - 1), 4, hipMemcpyDeviceToHost, pstream) != hipSuccess ||
                        hipStreamSynchronize(pstream) != hipSuccess) {
                        (void)hipGetLastError();
                        return fail(FTK_ERR_HIP, "cannot read the last row of a region");
                    }
                    complete = (long long)last >= reg_stop;
                }
                if (clk.on)
                    fprintf(stderr, "[ftk stream text] region %lld-%lld: piece %d ends at file offset %lld (hint %lld, contig ends %lld): %s\n",
                            reg_start, reg_stop, k, piece_off + (long long)n, read_end, hard_read_end,
                            complete ? "complete" : "a long row hides the end, reading on");
                if (complete) break;
                read_end = std::min(hard_read_end, read_end + (long long)(size_t(8) << 20));
            }
            const size_t raw_carry_d = n - used;
            if (piece_off >= 0) piece_off += (long long)used;
            // (with a copy of buf in flight the carried bytes are only read here: fill() moves them into the next buffer)
            if (raw_carry_d && !buf_in_flight) memmove(buf.data(), buf.data() + used, raw_carry_d);
            clk.lap(5);
            n = fill(buf, raw_carry_d, buf_in_flight ? used : 0);
            clk.lap(0);
            mark(k, "next piece read");
            eof = n - raw_carry_d < last_want;
            soft = at_soft_end(n);
            {
                std::lock_guard<std::mutex> lk(mu);
                if (stop) return false;
            }
            continue;
        }
        if (!submit_backs(k - 1, true)) return false;
        backs = k + 1;  // (this piece goes onto the parse stream right here)
        meta[k % kSets] = PieceMeta{};
        S.inflated = false;
        if (!S.ensure(carry + total + 2)) return fail(FTK_ERR_OOM, "out of page-locked / device memory for the text piece");
        if (carry) memcpy(S.h_text, carry_src, carry);
        clk.lap(5);
        if (!blocks.empty() && inflate_block_list(buf.data(), blocks, n_threads, S.h_text + carry, false, true) != FTK_OK)
            return fail(FTK_ERR_FORMAT, "BGZF inflate failed or block CRC mismatch");  // (CRCs checked like the GPU's pieces)
        clk.lap(1);
        char* b = (char*)S.h_text;
        char* e = b + carry + total;
        if (first_skip) {  // after an index seek: the contig starts inside the first block
            b += std::min<size_t>(first_skip, (size_t)(e - b));
            first_skip = 0;
        }
        if (!layout_known) {  // io/alignment.py:143-156: BED6 when the first data row has > 5 columns
            const char* q = b;
            while (q < e) {
                const char* nl = (const char*)memchr(q, '\n', (size_t)(e - q));
                const char* le = nl ? nl : e;
                if (le > q && *q != '#') {
                    int tabs = 0;
                    for (const char* x = q; x < le; ++x) tabs += (*x == '\t');
                    bed6 = (tabs + 1) > 5;
                    layout_known = true;
                    break;
                }
                if (!nl) break;
                q = nl + 1;
            }
        }
        char* last = e;
        if (!eof) {
            while (last > b && last[-1] != '\n') --last;
        } else if (e > b && e[-1] != '\n') {
            *e++ = '\n';  // the last row of the file has no line end: give it one (the buffer has the room)
            last = e;
        }
        if (last > b) {
            S.cut_tail = eof && partial_tail_ok;
            S.off = (size_t)(b - (char*)S.h_text);
            S.len = (size_t)(last - b);
            S.host_only = S.len >= (size_t(1) << 32) - 4096;
            bool ok = S.host_only || (hipMemsetAsync(S.d_sum, 0, sizeof(ftk::TextSummary), pstream) == hipSuccess &&
                                      hipMemcpyAsync(S.d_text, b, S.len, hipMemcpyHostToDevice, pstream) == hipSuccess);
            if (ok && !S.host_only) {
                ftk::textparse_launch(pstream, S.d_text, S.len, bed6, S.d_blocks, S.max_lines, S.d_s, S.d_e, S.d_q,
                                      S.d_t, S.d_sum);
                ok = hipGetLastError() == hipSuccess &&
                     hipMemcpyAsync(S.h_sum, S.d_sum, sizeof(ftk::TextSummary), hipMemcpyDeviceToHost, pstream) == hipSuccess &&
                     hipEventRecord(S.done, pstream) == hipSuccess;
            }
            if (!ok) {
                (void)hipGetLastError();
                return fail(FTK_ERR_HIP, "cannot launch the device row parser");
            }
            S.pending = true;
        }
        clk.lap(2);
        carry = (size_t)(e - last);
        carry_src = (const uint8_t*)last;
        if (!settle(k, false)) return false;
        clk.lap(3);
        if (eof) break;
        const size_t raw_carry = n - used;
        if (raw_carry) memmove(buf.data(), buf.data() + used, raw_carry);
        clk.lap(5);
        n = fill(buf, raw_carry);
        clk.lap(0);
        eof = n - raw_carry < last_want;
        {
            std::lock_guard<std::mutex> lk(mu);
            if (stop) return false;
        }
    }
    if (!submit_backs(k, true) || !settle(k, true)) return false;  // the pieces still in flight, in file order
    clk.lap(3);
    if (have_cur && !emit_device(std::move(cur))) return false;
    clk.lap(4);
    clk.report("text, device rows (parse = launch, merge = collect)");
    if (clk.on)
        fprintf(stderr, "[ftk stream text] %zu pieces parsed on the device, %zu by the host; %zu inflated by the host threads beside the GPU "
                        "(waited %.1f ms for them)\n", gpu_pieces, host_pieces, host_inflated, t_jobwait);
    if (clk.on && n_timed)
        fprintf(stderr, "[ftk stream text] device time per piece: front (copy up, inflate, CRC) avg %.2f max %.2f ms, back (set-up, rows) avg %.2f max %.2f ms, %zu pieces\n",
                front_ms / n_timed, front_max, back_ms / n_timed, back_max, n_timed);
    if (clk.on && atoi(getenv("FTK_DECODE_TIMING")) >= 2) fputs(trail.c_str(), stderr);
    return true;
}

bool ftk_fragstream::run_bam(RawBuf& buf, size_t n_first) {
    StageClock clk(this);
    size_t n_stretches = 0, n_redone = 0;
    RawBuf data;                // carry (partial record / header) + this piece's inflated bytes (host inflate)
    std::vector<uint8_t> carry_buf;  // the same carry while the pieces are inflated on the device
    size_t carry = 0;
    bool header_done = false;
    std::vector<int> wanted;    // ref id -> 1 when selected
    Contig cur;
    size_t cur_rows = 0;
    int cur_ref = -1;
    std::set<int> seen;
    size_t pending_skip = 0;
    // One piece of the file: compressed bytes in buf[0, n), its whole BGZF blocks, and (device inflate) the slot
    // that inflates it.
    struct Piece {
        size_t n = 0, used = 0, total = 0;
        bool eof = false, listed = false;
        std::vector<Block> blocks;
        int slot = -1;
    };
    // Device inflate (default on a stream that knows its GPU; FTK_DEVICE_INFLATE=0: host threads): the blocks of
    // pieces k+1 .. k+kAhead are inflated on the GPU and copied back while the host walks the records of piece k.
    // One slot and one HIP stream per piece in flight: a 48 MB piece holds ~800 blocks = 800 wavefronts, a
    // quarter of what the chip holds at this kernel's occupancy, and a block's decode chain takes the same ~6.5 ms
    // whether the chip is full or not - so three pieces' kernels (and their copies, either direction) run side by
    // side.  A slot's output is page-locked memory with room in front for the carry.
    constexpr size_t kRoom = size_t(32) << 20;
    constexpr int kAhead = 7, kSlots = kAhead + 1;
    static const bool want_dinf = !(getenv("FTK_DEVICE_INFLATE") && atoi(getenv("FTK_DEVICE_INFLATE")) == 0);
    const int device = inflate_device;  // (shadows the member: this path's GPU)
    bool dinf = want_dinf && device >= 0;
    FrontStreams<kSlots> streams;  // (slot 0 runs on the member pstream, which is destroyed with the stream object)
    if (dinf) {
        const bool ok = hipSetDevice(device) == hipSuccess && (pstream || (pstream = stream_pool().take(device)) != nullptr);
        if (!ok) {
            (void)hipGetLastError();
            dinf = false;
        }
        streams.device = device;
        streams.fallback = pstream;
        if (dinf) streams.prefill(std::min(pieces_expected(), kSlots) - 1);
    }
    DevSet sets[kSlots];
    if (dinf)
        for (auto& S : sets) S = devset_pool().take(device);
    struct Cleanup {
        DevSet* s;
        int device;
        hipStream_t pst;
        FrontStreams<kSlots>* streams;
        bool on;
        ~Cleanup() {
            if (!on) return;
            streams->settle_and_give();
            (void)hipStreamSynchronize(pst);
            for (int k = 0; k < kSlots; ++k) devset_pool().give(device, s[k]);
        }
    } cleanup{sets, device, pstream, &streams, dinf};
    // Every third piece of the look-ahead is inflated by the host threads instead (straight into its slot's
    // page-locked output, while the GPU works on the two in front of it): the chip turns a 64 KB block of BAM over
    // every 2.7 us = 24 GB/s of records, the 16 threads manage 11 GB/s, and between record walks they have nothing
    // else to do.  FTK_BAM_HOST_SHARE=<n>: every n-th piece (0: none).
    static const int host_share = [] {
        const char* e = getenv("FTK_BAM_HOST_SHARE");
        return e ? atoi(e) : 3;
    }();
    bool slot_on_host[kSlots] = {};
    std::future<int> host_job[kSlots];  // the host threads' inflate of a slot's piece (from the slot's own copy of the bytes)
    struct JobGuard {  // no job outlives the buffers it works on
        std::future<int>* j;
        ~JobGuard() {
            for (int k = 0; k < kSlots; ++k)
                if (j[k].valid()) (void)j[k].get();
        }
    } job_guard{host_job};
    auto submit = [&](Piece& pc, int index) -> bool {  // index: the piece's number among the submitted ones
        const int slot = index % kSlots;
        DevSet& S = sets[slot];
        hipStream_t pstream = slot ? streams.get(slot) : this->pstream;  // (shadows the member: this slot's stream)
        if (pc.total + kRoom + 64 >= (size_t(1) << 32)) return fail(FTK_ERR_FORMAT, "BGZF piece too large");
        slot_on_host[slot] = host_share > 0 && header_done && (index % host_share) == host_share - 1;
        if (slot_on_host[slot]) {
            if (!S.ensure(kRoom + pc.total + 64) || !S.ensure_host_comp(pc.used + 64))
                return fail(FTK_ERR_OOM, "out of page-locked memory for the BAM piece");
            {
                const size_t used = pc.used;
                const int nt = std::max(1, std::min(n_threads, (int)(used >> 20) + 1));
                const uint8_t* src = buf.data();
                uint8_t* dst = S.h_comp;
                parallel_run(nt, [&](int t) {
                    const size_t a = used * (size_t)t / nt, b2 = used * (size_t)(t + 1) / nt;
                    memcpy(dst + a, src + a, b2 - a);
                });
            }
            // (the job owns its block list; the bytes stay in the slot until the piece has been walked)
            host_job[slot] = std::async(std::launch::async, [blocks = pc.blocks, comp = (const uint8_t*)S.h_comp,
                                                             out = S.h_text + kRoom, nt = std::max(1, n_threads - 2)] {
                return blocks.empty() ? (int)FTK_OK : inflate_block_list(comp, blocks, nt, out, true, true);  // CRCs checked like the GPU's pieces
            });
            pc.slot = slot;
            return true;
        }
        if (!S.ensure(kRoom + pc.total + 64) || !S.ensure_inflate(pc.used, pc.blocks.size()) || !S.ensure_host_comp(pc.used + 64))
            return fail(FTK_ERR_OOM, "out of page-locked / device memory for the BAM piece");
        {
            const size_t used = pc.used;
            const int nt = std::max(1, std::min(n_threads, (int)(used >> 20) + 1));
            const uint8_t* src = buf.data();
            uint8_t* dst = S.h_comp;
            parallel_run(nt, [&](int t) {
                const size_t a = used * (size_t)t / nt, b2 = used * (size_t)(t + 1) / nt;
                memcpy(dst + a, src + a, b2 - a);
            });
        }
        for (size_t i = 0; i < pc.blocks.size(); ++i) {
            const Block& bl = pc.blocks[i];
            S.h_tab[i] = {(uint32_t)bl.in_off, (uint32_t)bl.in_len, (uint32_t)(kRoom + bl.out_off), (uint32_t)bl.out_len};
            const uint8_t* tr = buf.data() + bl.in_off + bl.in_len;
            S.want_crc[i] = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
        }
        S.n_tab = pc.blocks.size();
        bool ok = hipMemsetAsync(S.d_ist, 0, sizeof(ftk::InflateStatus), pstream) == hipSuccess &&
                  (pc.used == 0 || hipMemcpyAsync(S.d_comp, S.h_comp, pc.used, hipMemcpyHostToDevice, pstream) == hipSuccess) &&
                  (pc.blocks.empty() || hipMemcpyAsync(S.d_tab, S.h_tab, pc.blocks.size() * sizeof(ftk::InflateBlock),
                                                       hipMemcpyHostToDevice, pstream) == hipSuccess);
        if (ok) {
            ftk::inflate_launch(pstream, S.d_comp, S.d_tab, (int)pc.blocks.size(), S.d_text, S.d_ist, S.d_crc, /*vector_matches=*/true);
            ok = hipGetLastError() == hipSuccess &&
                 (pc.total == 0 || hipMemcpyAsync(S.h_text + kRoom, S.d_text + kRoom, pc.total, hipMemcpyDeviceToHost, pstream) == hipSuccess) &&
                 hipMemcpyAsync(S.h_ist, S.d_ist, sizeof(ftk::InflateStatus), hipMemcpyDeviceToHost, pstream) == hipSuccess &&
                 (pc.blocks.empty() || hipMemcpyAsync(S.h_crc, S.d_crc, pc.blocks.size() * 4, hipMemcpyDeviceToHost, pstream) == hipSuccess) &&
                 hipEventRecord(S.done, pstream) == hipSuccess;
        }
        if (!ok) {
            (void)hipGetLastError();
            return fail(FTK_ERR_HIP, "cannot launch the device inflate");
        }
        pc.slot = slot;
        return true;
    };
    auto wait_slot = [&](int slot) -> bool {
        DevSet& S = sets[slot];
        if (slot_on_host[slot]) {  // inflated by the host threads
            if (host_job[slot].valid() && host_job[slot].get() != FTK_OK)
                return fail(FTK_ERR_FORMAT, "BGZF inflate failed or block CRC mismatch (host share of a device stream)");
            return true;
        }
        if (hipEventSynchronize(S.done) != hipSuccess) {
            (void)hipGetLastError();
            return fail(FTK_ERR_HIP, "the device inflate failed");
        }
        if (S.h_ist->n_bad) return fail(FTK_ERR_FORMAT, "BGZF inflate failed");
        for (size_t i = 0; i < S.n_tab; ++i)
            if (S.h_crc[i] != S.want_crc[i]) return fail(FTK_ERR_FORMAT, "BGZF block CRC mismatch (device inflate)");
        return true;
    };
    auto list_blocks = [&](Piece& pc) -> bool {
        if (pc.listed) return true;
        if (!whole_blocks(buf.data(), pc.n, pc.eof, &pc.blocks, &pc.used, &pc.total)) return fail(FTK_ERR_FORMAT, "corrupt BGZF block");
        pc.listed = true;
        return true;
    };
    Piece curp;
    std::deque<Piece> ahead;  // pieces behind curp that are already on the device, in file order
    curp.n = n_first;
    curp.eof = n_first < last_want;
    int n_submitted = 0;
    for (;;) {
        if (!list_blocks(curp)) return false;
        if (dinf && curp.slot < 0 && !submit(curp, n_submitted++)) return false;
        // read the next pieces and start their inflate before this piece's records are walked (not while the header
        // is still being probed: an index seek may throw those reads away).  `buf` holds the compressed bytes of the
        // piece read last - the newest of `ahead`, or curp.
        if (dinf && header_done) {
            while ((int)ahead.size() < kAhead) {
                const Piece& last = ahead.empty() ? curp : ahead.back();
                if (last.eof) break;
                const size_t raw_carry = last.n - last.used;
                if (raw_carry) memmove(buf.data(), buf.data() + last.used, raw_carry);
                clk.lap(5);
                Piece np;
                np.n = fill(buf, raw_carry);
                clk.lap(0);
                np.eof = np.n - raw_carry < last_want;
                if (!list_blocks(np) || !submit(np, n_submitted++)) return false;
                ahead.push_back(std::move(np));
            }
        }
        size_t& n = curp.n;
        bool& eof = curp.eof;
        const size_t used = curp.used, total = curp.total;
        const uint8_t* p;
        clk.lap(5);
        if (dinf) {
            if (carry > kRoom) return fail(FTK_ERR_FORMAT, "a BAM header or record of more than 32 MB: set FTK_DEVICE_INFLATE=0");
            if (!wait_slot(curp.slot)) return false;
            uint8_t* base = sets[curp.slot].h_text + kRoom - carry;
            if (carry) memcpy(base, carry_buf.data(), carry);
            p = base;
        } else {
            if (!data.reserve(carry + total + 1)) return fail(FTK_ERR_OOM, "out of host memory");
            if (!curp.blocks.empty() && inflate_block_list(buf.data(), curp.blocks, n_threads, data.data() + carry) != FTK_OK)
                return fail(FTK_ERR_FORMAT, "BGZF inflate failed");
            p = data.data();
        }
        clk.lap(1);
        const size_t m = carry + total;
        size_t off = std::min(pending_skip, m);  // (a damaged index may point past the block)
        pending_skip = 0;
        if (!header_done) {
            bool complete = false;
            do {
                if (m < 12) break;
                if (memcmp(p, "BAM\1", 4) != 0) return fail(FTK_ERR_FORMAT, (path + " is not a BAM file").c_str());
                size_t o = 4;
                const uint32_t l_text = rd_u32(p + o);
                o += 4 + (size_t)l_text;
                if (o + 4 > m) break;
                const uint32_t n_ref = rd_u32(p + o);
                o += 4;
                std::vector<std::string> names;
                std::vector<int64_t> lens;
                bool cut = false;
                for (uint32_t r = 0; r < n_ref; ++r) {
                    if (o + 4 > m) { cut = true; break; }
                    const uint32_t l_name = rd_u32(p + o);
                    o += 4;
                    if (l_name == 0) return fail(FTK_ERR_FORMAT, "corrupt BAM reference list");
                    if (o + l_name + 4 > m) { cut = true; break; }
                    names.emplace_back((const char*)p + o, l_name - 1);
                    o += l_name;
                    lens.push_back(rd_i32(p + o));
                    o += 4;
                }
                if (cut) break;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    ref_names = names;
                    ref_lens = lens;
                    header_ready = true;
                    cv.notify_all();
                }
                wanted.assign(n_ref, 0);
                for (uint32_t r = 0; r < n_ref; ++r) wanted[r] = !has_only || names[r] == only;
                off = o;
                complete = true;
            } while (false);
            if (!complete) {
                if (eof) return fail(FTK_ERR_FORMAT, "truncated BAM header");
                carry = m;  // need more bytes: keep everything
                if (dinf) carry_buf.assign(p, p + m);
                goto next_piece;
            }
            header_done = true;
            ahead_ok = true;
            if (has_only) {  // BAI: jump to the contig's records instead of walking the whole file
                int target = -1;
                for (size_t r = 0; r < ref_names.size(); ++r)
                    if (ref_names[r] == only) target = (int)r;
                if (target < 0) return true;  // not in the header: nothing to hand out
                const IndexSpan sp = index_lookup(index_path_of(path, true), true, std::string(), target);
                if (sp.usable && !sp.present) return true;  // no alignment on this contig
                if (sp.usable && seek_to(sp)) {
                    carry = 0;
                    curp = Piece{};
                    curp.n = fill(buf, 0);
                    curp.eof = curp.n < last_want;
                    pending_skip = first_skip;
                    first_skip = 0;
                    continue;
                }
                read_end = -1;
                partial_tail_ok = false;
                first_skip = 0;
            }
        }
        {
            // The records form a chain (each starts where the previous one ends), and walking it is one
            // dependent cache miss per record - ~80 ns x millions.  So the piece is cut into byte ranges:
            // every thread GUESSES the first record start in its range (header plausibility, three links
            // deep), walks and parses from there; afterwards the chain is checked range by range - a range
            // whose guessed start is not where the previous range's walk ended is redone from the true
            // offset.  The result never depends on the guess, only the speed does.
            const int n_ref = (int)wanted.size();
            const int nseg = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, (m - off) / kBamStretch + 1));
            struct Stretch {
                size_t start = SIZE_MAX, landing = 0;
                bool bad = false;
                std::vector<BamRun> runs;
            };
            std::vector<Stretch> seg(nseg);
            auto walk = [&](size_t from, size_t until, Stretch& st) {
                st.runs.clear();
                st.bad = false;
                st.start = from;
                BamRun* run = nullptr;
                size_t o = from;
                while (o < until && o + 4 <= m) {
                    const uint32_t bs = rd_u32(p + o);
                    if (bs < 32) { st.bad = true; break; }
                    if (o + 4 + (size_t)bs > m) break;  // incomplete: waits for the next piece
                    const uint8_t* r = p + o + 4;
                    const int32_t ref_id = rd_i32(r);
                    if (ref_id >= 0 && ref_id < n_ref && wanted[ref_id]) {
                        if (!run || run->ref != ref_id) {
                            st.runs.push_back(BamRun{ref_id, {}});
                            run = &st.runs.back();
                            // both mates of a pair are at least ~150 bytes each: room for the rest of the stretch
                            const size_t guess = (until > o ? until - o : 0) / 300 + 16;
                            run->c.start.reserve(guess);
                            run->c.end.reserve(guess);
                            run->c.mapq.reserve(guess);
                            run->c.strand.reserve(guess);
                            run->c.r1s.reserve(guess);
                            run->c.r1e.reserve(guess);
                        }
                        bam_record(r, bs, run->c);
                    }
                    o += 4 + (size_t)bs;
                }
                st.landing = o;
            };
            auto bound = [&](int k) { return k >= nseg ? m : off + (m - off) * (size_t)k / (size_t)nseg; };
            parallel_run(nseg, [&](int k) {
                const size_t from = k == 0 ? off : guess_record_start(p, bound(k), m, n_ref);
                if (from == SIZE_MAX) return;  // nothing that looks like a record: settled by the check below
                walk(from, bound(k + 1), seg[k]);
            });
            clk.lap(2);
            size_t o = off;
            for (int k = 0; k < nseg; ++k) {
                ++n_stretches;
                if (seg[k].start != o || seg[k].bad) {
                    ++n_redone;
                    if (clk.on && getenv("FTK_DECODE_TIMING")[0] >= '2')
                        fprintf(stderr, "[ftk stream bam] stretch %d of %d redone: guessed %zd, chain at %zu, bound %zu, piece %zu bytes%s\n", k, nseg,
                                seg[k].start == SIZE_MAX ? (ssize_t)-1 : (ssize_t)seg[k].start, o, bound(k), m, seg[k].bad ? " (bad)" : "");
                    walk(o, bound(k + 1), seg[k]);
                    if (seg[k].bad) return fail(FTK_ERR_FORMAT, "corrupt BAM record");
                }
                o = seg[k].landing;
            }
            if (eof && o != m && !partial_tail_ok) return fail(FTK_ERR_FORMAT, "truncated BAM record");
            clk.lap(5);  // "other" holds the chain check (and any redone range)
            for (auto& st : seg)
                for (auto& r : st.runs) {
                    if (emitted_refs.count(r.ref)) continue;  // handed out by the device pass this one replaces
                    if (cur_ref >= 0 && r.ref != cur_ref) {
                        clk.lap(3);
                        if (cur_rows && !emit(std::move(cur))) return false;
                        clk.lap(4);
                        cur = Contig{};
                        cur_rows = 0;
                        cur_ref = -1;
                    }
                    if (cur_ref < 0) {
                        if (!seen.insert(r.ref).second)
                            return fail(FTK_ERR_UNSORTED, ("contig " + ref_names[r.ref] + " appears in two separate runs: the BAM is not coordinate-sorted").c_str());
                        cur_ref = r.ref;
                        cur.name = ref_names[r.ref];
                        cur.length = ref_lens[r.ref];
                    }
                    cur_rows += r.c.start.size();
                    skipped[0] += r.c.skipped[0];
                    skipped[1] += r.c.skipped[1];
                    if (!r.c.start.empty()) cur.parts.push_back(std::move(r.c));  // sorted / gathered by the packer
                }
            clk.lap(3);
            carry = m - o;
            if (dinf) carry_buf.assign(p + o, p + o + carry);
            else if (carry) memmove(data.data(), p + o, carry);
        }
    next_piece:
        if (eof) break;
        if (!ahead.empty()) {
            curp = std::move(ahead.front());
            ahead.pop_front();
        } else {
            const size_t raw_carry = n - used;
            if (raw_carry) memmove(buf.data(), buf.data() + used, raw_carry);
            clk.lap(5);
            Piece np;
            np.n = fill(buf, raw_carry);
            clk.lap(0);
            np.eof = np.n - raw_carry < last_want;
            curp = std::move(np);
        }
        {
            std::lock_guard<std::mutex> lk(mu);
            if (stop) return false;
        }
    }
    clk.lap(5);
    if (cur_ref >= 0 && cur_rows && !emit(std::move(cur))) return false;
    clk.lap(4);
    clk.report("bam");
    if (clk.on) fprintf(stderr, "[ftk stream bam] %zu stretches of the record chain, %zu redone after the chain check\n", n_stretches, n_redone);
    return true;
}

// Sorted, device-resident BAM contig -> the consumer (the counterpart of emit_device for text contigs).
bool ftk_fragstream::emit_device_bam(Contig&& ct) {
    DevColumns& d = *ct.dev;
    if (hipEventCreateWithFlags(&d.ready, hipEventDisableTiming) != hipSuccess || hipEventRecord(d.ready, pstream) != hipSuccess) {
        (void)hipGetLastError();
        return fail(FTK_ERR_HIP, "cannot record the contig's ready event");
    }
    std::unique_ptr<ftk_fragtable> t(new ftk_fragtable());
    t->bam = true;
    ct.p.rows = d.rows;
    ct.p.start = d.start;
    ct.p.end = d.end;
    ct.p.mapq = d.mapq;
    ct.p.strand = d.strand;
    ct.p.r1s = d.r1s;
    ct.p.r1e = d.r1e;
    ct.p.ord = d.ord;
    t->contigs.push_back(std::move(ct));
    std::unique_lock<std::mutex> lk(mu);
    cv.wait(lk, [&] { return stop || ready.size() < max_queued; });
    if (stop) return false;
    ready.push_back(t.release());
    cv.notify_all();
    return true;
}

bool ftk_fragstream::run_bam_device(RawBuf& buf, size_t n_first) {
    StageClock clk(this);
    const int device = inflate_device;
    constexpr size_t kRoom = size_t(32) << 20;
    constexpr int kAhead = 7, kSlots = kAhead + 1;  // (3 ahead: the 60x slice 0.050 s; 5: 0.038-0.044; 7: 0.036-0.042)
    // bytes of records per thread of the chain walk (FTK_BAM_DEV_STRETCH: tests walk tiny stretches).  A thread follows
    // its stretch's records link by link - dependent loads - so shorter stretches are shorter chains on more threads,
    // until the guesses at their starts and the fix passes cost more than they save: the 5.9 GB BAM, warm passes
    // alternated on one box (tools/env_ab.sh), 32 KB 0.284-0.300 s, 16 KB (rounds 3-4) 0.253-0.287, 8 KB 0.248-0.269,
    // 4 KB 0.250-0.270, 2 KB 0.254-0.268, 1 KB 0.283-0.314
    static const uint32_t stretch_bytes = [] {
        const char* e = getenv("FTK_BAM_DEV_STRETCH");
        const long v = e ? atol(e) : 0;
        return (uint32_t)(v >= 64 ? v : 8192);
    }();
    if (hipSetDevice(device) != hipSuccess || (!pstream && (pstream = stream_pool().take(device)) == nullptr)) {
        (void)hipGetLastError();
        return fail(FTK_ERR_HIP, "cannot create the parse stream");
    }
    FrontStreams<kSlots> streams;
    streams.device = device;
    streams.fallback = pstream;
    streams.prefill(std::min(pieces_expected(), kSlots) - 1);
    DevSet sets[kSlots];
    for (auto& S : sets) S = devset_pool().take(device);
    uint8_t* d_wanted = nullptr;
    struct Cleanup {
        DevSet* s;
        int device;
        hipStream_t pst;
        FrontStreams<kSlots>* streams;
        uint8_t** wanted;
        ~Cleanup() {
            streams->settle_and_give();
            (void)hipStreamSynchronize(pst);
            for (int k = 0; k < kSlots; ++k) devset_pool().give(device, s[k]);
            if (*wanted) (void)hipFree(*wanted);
        }
    } cleanup{sets, device, pstream, &streams, &d_wanted};

    struct Piece {
        size_t n = 0, used = 0, total = 0;
        bool eof = false;
        std::vector<Block> blocks;
        int slot = -1;
        uint32_t first_off = 0;
        bool has_prev = false;
        int prev_slot = -1;
        long long file_off = -1; // where the piece's first byte lies in the file (-1: unknown)
        bool on_host = false;    // inflated by the host threads (its text goes up before the parse)
        bool back_done = false;  // the parse is on the parse stream
    };
    // Every third piece is inflated by the host threads beside the GPU (the chip turns BAM blocks over at ~24 GB/s,
    // the 16 threads manage ~11 GB/s and have nothing else to do now that the records stay on the device); its text
    // goes up in one DMA before its parse.  FTK_BAM_HOST_SHARE=<n>: every n-th piece (0: none).
    // Only with enough threads to keep up: a rank of eight on a 16-core quota has two, and a piece that took 10 ms
    // on fourteen threads would hold the pipeline for 70.
    static const int host_share_env = [] {
        const char* e = getenv("FTK_BAM_HOST_SHARE");
        return e ? atoi(e) : -1;
    }();
    // (round 5, with the lane-parallel inflate loop - tools/env_ab.sh bam, five runs each on one box, warm passes of the
    // 5.9 GB file: none 0.238-0.251 s, every 32nd 0.221-0.241, every 16th 0.214-0.225, 12th 0.253-0.258, 8th 0.251-0.281,
    // 6th - the default of round 4 - 0.234-0.249, 4th 0.31-0.33)
    const int host_share = host_share_env >= 0 ? host_share_env : (n_threads >= 8 ? 16 : 0);
    std::future<int> host_job[kSlots];
    struct JobGuard {  // no job outlives the buffers it works on
        std::future<int>* j;
        ~JobGuard() {
            for (int k = 0; k < kSlots; ++k)
                if (j[k].valid()) (void)j[k].get();
        }
    } job_guard{host_job};
    // (a GPU piece's compressed bytes go up straight from the page-locked read buffer; fill() lets that buffer rest
    // until the copy is done - see buf_in_flight)
    struct RestGuard {
        ftk_fragstream* s;
        ~RestGuard() { s->drop_resting(); }
    } rest_guard{this};
    double t_jobwait = 0, t_front = 0, t_header = 0;  // FTK_DECODE_TIMING: what "other" is made of
    auto tick = [] { return std::chrono::steady_clock::now(); };
    auto since = [](std::chrono::steady_clock::time_point t0) {
        return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    };
    auto list_blocks = [&](Piece& pc) -> bool {
        if (!whole_blocks(buf.data(), pc.n, pc.eof, &pc.blocks, &pc.used, &pc.total)) return fail(FTK_ERR_FORMAT, "corrupt BGZF block");
        return true;
    };
    // front of a piece (its slot's own stream): compressed bytes up, inflate, CRC
    auto submit_front = [&](Piece& pc, int index) -> bool {
        const auto t_in = tick();
        struct Acc { double* d; std::chrono::steady_clock::time_point t0; ~Acc() { *d += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); } } acc{&t_front, t_in};
        const int slot = index % kSlots;
        DevSet& S = sets[slot];
        hipStream_t st = streams.get(slot);
        if (S.pending) return fail(FTK_ERR_HIP, "buffer ring out of step");
        // The doubled pieces of a large BAM (run_guarded) assume records that deflate 3-5 x.  The first pieces of a
        // stream are short (the read ramp): one that inflates 16 x or more says the file is of another kind (synthetic,
        // all-N reads), and the stream goes on with the standard pieces before a 96 MB piece can outgrow the 4 GiB a
        // piece's text may take.
        if (index == 0 && piece_bytes > kStreamPiece && pc.used > 0 && pc.total / pc.used >= 16) piece_bytes = kStreamPiece;
        if (pc.total + kRoom + 64 >= (size_t(1) << 32)) {
            if (piece_bytes > kStreamPiece) {
                // (a file whose compression rises behind its first piece: the HOST decoder takes it from the start,
                // skipping the contigs handed out - slower, correct, and not met on any file so far)
                piece_bytes = kStreamPiece;
                want_host_restart = true;
                return false;
            }
            return fail(FTK_ERR_FORMAT, "BGZF piece too large");
        }
        pc.on_host = host_share > 0 && index > 0 && (index % host_share) == host_share - 1;
        // a GPU piece goes up straight from the (page-locked) read buffer; the host threads work on their own copy of
        // theirs, which they read from the file (page cache) themselves when the piece's file offset is known
        hipEvent_t up_ev = (buf.pinned && !pc.on_host) ? take_up_event() : nullptr;
        const bool direct = up_ev != nullptr;
        const bool job_reads = pc.on_host && pc.file_off >= 0;
        if (!S.ensure(kRoom + pc.total + 64, false) || (pc.on_host && !S.ensure_host_text(kRoom + pc.total + 64)) ||
            !S.ensure_inflate(pc.used, pc.blocks.size()) || (!direct && !S.ensure_host_comp(pc.used + 64)) ||
            !S.ensure_bam(kRoom + pc.total + 64, stretch_bytes))
            return fail(FTK_ERR_OOM, "out of page-locked / device memory for the BAM piece");
        if (!direct && !job_reads) {
            const size_t used = pc.used;
            const int nt = std::max(1, std::min(n_threads, (int)(used >> 20) + 1));
            const uint8_t* src = buf.data();
            uint8_t* dst = S.h_comp;
            parallel_run(nt, [&](int t) {
                const size_t a = used * (size_t)t / nt, b2 = used * (size_t)(t + 1) / nt;
                memcpy(dst + a, src + a, b2 - a);
            });
        }
        for (size_t i = 0; i < pc.blocks.size(); ++i) {
            const Block& bl = pc.blocks[i];
            S.h_tab[i] = {(uint32_t)bl.in_off, (uint32_t)bl.in_len, (uint32_t)(kRoom + bl.out_off), (uint32_t)bl.out_len};
            const uint8_t* tr = buf.data() + bl.in_off + bl.in_len;
            S.want_crc[i] = (uint32_t)tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
        }
        S.n_tab = pc.blocks.size();
        if (pc.on_host) {
            // (the job owns its block list; compressed bytes and output are the slot's page-locked buffers)
            host_job[slot] = std::async(std::launch::async, [blocks = pc.blocks, comp = S.h_comp, out = S.h_text + kRoom,
                                                             nt = std::max(1, n_threads - 2), fd = fileno(fp), used = pc.used,
                                                             off = pc.file_off, job_reads] {
                if (job_reads) {  // its own copy of the compressed bytes, four pread threads
                    std::atomic<int> bad{0};
                    std::vector<std::thread> th;
                    auto part = [&](int t) {
                        size_t a = used * (size_t)t / 4;
                        const size_t e = used * (size_t)(t + 1) / 4;
                        while (a < e) {
                            const ssize_t r = pread(fd, comp + a, e - a, (off_t)(off + (long long)a));
                            if (r <= 0) { bad.store(1); return; }
                            a += (size_t)r;
                        }
                    };
                    for (int t = 1; t < 4; ++t) th.emplace_back(part, t);
                    part(0);
                    for (auto& t : th) t.join();
                    if (bad.load()) return (int)FTK_ERR_IO;
                }
                return blocks.empty() ? (int)FTK_OK : inflate_block_list(comp, blocks, nt, out, true, true);  // CRCs checked
            });
            pc.slot = slot;
            return true;
        }
        // (the compressed bytes go up at once - nothing of the slot's previous piece uses d_comp any more - and only
        // the inflate, which overwrites the text the appends may still read, waits for the slot's release)
        bool ok = (pc.used == 0 || hipMemcpyAsync(S.d_comp, direct ? buf.data() : S.h_comp, pc.used, hipMemcpyHostToDevice, st) == hipSuccess) &&
                  (!direct || hipEventRecord(up_ev, st) == hipSuccess) &&
                  (!S.freed_valid || hipStreamWaitEvent(st, S.freed, 0) == hipSuccess) &&
                  hipMemsetAsync(S.d_ist, 0, sizeof(ftk::InflateStatus), st) == hipSuccess &&
                  (pc.blocks.empty() || hipMemcpyAsync(S.d_tab, S.h_tab, pc.blocks.size() * sizeof(ftk::InflateBlock),
                                                       hipMemcpyHostToDevice, st) == hipSuccess);
        if (direct) buf_in_flight = up_ev;  // (fill() parks the buffer behind it)
        if (ok) {
            ftk::inflate_launch(st, S.d_comp, S.d_tab, (int)pc.blocks.size(), S.d_text, S.d_ist, S.d_crc, /*vector_matches=*/true);
            ok = hipGetLastError() == hipSuccess && hipEventRecord(S.front, st) == hipSuccess;
        }
        if (!ok) {
            (void)hipGetLastError();
            return fail(FTK_ERR_HIP, "cannot launch the device inflate");
        }
        pc.slot = slot;
        return true;
    };
    // back of a piece (the parse stream, piece after piece): the record chain behind the previous piece's
    int n_ref = 0;
    auto submit_back = [&](Piece& pc) -> bool {
        DevSet& S = sets[pc.slot];
        DevSet* P = pc.has_prev ? &sets[pc.prev_slot] : nullptr;
        if (pc.on_host) {  // the host threads' text goes up on the slot's stream (behind the appends that read the set last)
            hipStream_t st = streams.get(pc.slot);
            const auto t0 = tick();
            const int jrc = host_job[pc.slot].valid() ? host_job[pc.slot].get() : (int)FTK_OK;
            t_jobwait += since(t0);
            if (jrc != FTK_OK)
                return fail(FTK_ERR_FORMAT, "BGZF inflate failed or block CRC mismatch (host share of a device stream)");
            const bool up = (!S.freed_valid || hipStreamWaitEvent(st, S.freed, 0) == hipSuccess) &&
                            hipMemsetAsync(S.d_ist, 0, sizeof(ftk::InflateStatus), st) == hipSuccess &&
                            (pc.total == 0 || hipMemcpyAsync(S.d_text + kRoom, S.h_text + kRoom, pc.total, hipMemcpyHostToDevice, st) == hipSuccess) &&
                            hipEventRecord(S.front, st) == hipSuccess;
            if (!up) {
                (void)hipGetLastError();
                return fail(FTK_ERR_HIP, "cannot send a host-inflated piece to the device");
            }
            S.n_tab = 0;  // (its CRCs were checked by the host job)
        }
        bool ok = hipMemsetAsync(S.d_bsum, 0, sizeof(ftk::BamSummary), pstream) == hipSuccess &&
                  hipStreamWaitEvent(pstream, S.front, 0) == hipSuccess;
        if (ok) {
            ftk::bamparse_launch(pstream, S.d_text, (uint32_t)kRoom, (uint32_t)pc.total, P ? P->d_text : nullptr, P ? P->d_bsum : nullptr,
                                 pc.first_off, d_wanted, n_ref, stretch_bytes, S.d_stretch, S.stretch_words, S.max_lines, S.d_s,
                                 S.d_e, S.d_q, S.d_t, S.d_r1s, S.d_r1e, S.d_ref, S.d_bsum);
            ok = hipGetLastError() == hipSuccess &&
                 hipMemcpyAsync(S.h_bsum, S.d_bsum, sizeof(ftk::BamSummary), hipMemcpyDeviceToHost, pstream) == hipSuccess &&
                 hipMemcpyAsync(S.h_ist, S.d_ist, sizeof(ftk::InflateStatus), hipMemcpyDeviceToHost, pstream) == hipSuccess &&
                 (S.n_tab == 0 || hipMemcpyAsync(S.h_crc, S.d_crc, S.n_tab * 4, hipMemcpyDeviceToHost, pstream) == hipSuccess) &&
                 hipEventRecord(S.done, pstream) == hipSuccess;
        }
        if (!ok) {
            (void)hipGetLastError();
            return fail(FTK_ERR_HIP, "cannot launch the device record parser");
        }
        if (P) {  // the previous piece's text has been read (the carry): its slot may take a new piece behind this point
            P->freed_valid = hipEventRecord(P->freed, pstream) == hipSuccess;
            if (!P->freed_valid) {
                (void)hipGetLastError();
                return fail(FTK_ERR_HIP, "cannot record a buffer set's release");
            }
        }
        S.pending = true;
        pc.back_done = true;
        return true;
    };

    // ---- the header: the first piece is inflated on the device, its head copied back until the header is complete
    int region_ref = -1;                // (region reads) the reference the region lies on
    unsigned long long last_key = 0;    // the last record settled so far: (reference << 32) | position
    Piece curp;
    curp.n = n_first;
    curp.eof = n_first < last_want;
    curp.file_off = 0;  // (run_guarded read the first piece from the start of the file)
    int n_submitted = 0;
    if (!list_blocks(curp)) return false;
    std::deque<Piece> ahead;  // pieces behind curp whose fronts (and, in file order, backs) are enqueued
    auto read_ahead_fronts = [&]() -> bool {  // read the next pieces and start their inflate
        while ((int)ahead.size() < kAhead) {
            const Piece& last = ahead.empty() ? curp : ahead.back();
            if (last.eof) break;
            const size_t raw_carry = last.n - last.used;
            // (with a copy of buf in flight the carried bytes are only read: fill() moves them into the next buffer)
            const size_t carry_at = buf_in_flight ? last.used : 0;
            if (raw_carry && !buf_in_flight) memmove(buf.data(), buf.data() + last.used, raw_carry);
            clk.lap(5);
            Piece np;
            np.n = fill(buf, raw_carry, carry_at);
            clk.lap(0);
            np.eof = np.n - raw_carry < last_want;
            np.has_prev = true;
            np.prev_slot = last.slot;
            np.file_off = last.file_off >= 0 ? last.file_off + (long long)last.used : -1;
            if (!list_blocks(np) || !submit_front(np, n_submitted++)) return false;
            ahead.push_back(std::move(np));
        }
        return true;
    };

    std::vector<int> wanted;
    const auto t_head0 = tick();
    {
        // The header sits in the first BGZF blocks: those are inflated right here on the host (a few blocks of 64 KB,
        // microseconds) so that the pipeline does not wait for the first piece's trip through the device.
        std::vector<uint8_t> head;
        size_t nb = 0, o = 0;
        bool complete = false;
        while (!complete) {
            const size_t take = std::min(curp.blocks.size(), std::max<size_t>(2 * nb, 4));
            if (take == nb) {  // the header does not end inside the first piece: the host path
                if (curp.eof) return fail(FTK_ERR_FORMAT, "truncated BAM header");
                want_host_restart = true;
                return false;
            }
            nb = take;
            std::vector<Block> first(curp.blocks.begin(), curp.blocks.begin() + nb);
            const size_t m = first.back().out_off + first.back().out_len;
            head.resize(m + 1);
            if (inflate_block_list(buf.data(), first, std::min(n_threads, 4), head.data()) != FTK_OK)
                return fail(FTK_ERR_FORMAT, "BGZF inflate failed");
            const uint8_t* p = head.data();
            do {
                if (m < 12) break;
                if (memcmp(p, "BAM\1", 4) != 0) return fail(FTK_ERR_FORMAT, (path + " is not a BAM file").c_str());
                o = 4;
                const uint32_t l_text = rd_u32(p + o);
                o += 4 + (size_t)l_text;
                if (o + 4 > m) break;
                const uint32_t nr = rd_u32(p + o);
                o += 4;
                std::vector<std::string> names;
                std::vector<int64_t> lens;
                bool cut = false;
                for (uint32_t r = 0; r < nr; ++r) {
                    if (o + 4 > m) { cut = true; break; }
                    const uint32_t l_name = rd_u32(p + o);
                    o += 4;
                    if (l_name == 0) return fail(FTK_ERR_FORMAT, "corrupt BAM reference list");
                    if (o + l_name + 4 > m) { cut = true; break; }
                    names.emplace_back((const char*)p + o, l_name - 1);
                    o += l_name;
                    lens.push_back(rd_i32(p + o));
                    o += 4;
                }
                if (cut) break;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    ref_names = names;
                    ref_lens = lens;
                    header_ready = true;
                    cv.notify_all();
                }
                wanted.assign(nr, 0);
                for (uint32_t r = 0; r < nr; ++r) wanted[r] = !has_only || names[r] == only;
                complete = true;
            } while (false);
        }
        n_ref = (int)wanted.size();
        std::vector<uint8_t> w8(std::max<size_t>(wanted.size(), 1), 0);
        for (size_t r = 0; r < wanted.size(); ++r) w8[r] = (uint8_t)wanted[r];
        if (hipMalloc((void**)&d_wanted, w8.size()) != hipSuccess ||
            hipMemcpy(d_wanted, w8.data(), w8.size(), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipGetLastError();
            return fail(FTK_ERR_OOM, "out of device memory");
        }
        curp.first_off = (uint32_t)o;
        ahead_ok = true;
        if (has_only) {  // BAI: jump to the contig's records instead of walking the whole file
            int target = -1;
            for (size_t r = 0; r < ref_names.size(); ++r)
                if (ref_names[r] == only) target = (int)r;
            if (target < 0) return true;
            IndexSpan sp = index_lookup(index_path_of(path, true), true, std::string(), target, has_region ? reg_start : -1,
                                        has_region ? reg_stop : -1);
            if (sp.usable && !sp.present) return true;
            if (has_region && !(sp.usable && sp.region)) has_region = false;  // (the whole contig: a superset)
            if (has_region) sp.beg = sp.reg_beg;  // start at the first record that overlaps the region ...
            region_ref = target;
            if (sp.usable && seek_to(sp)) {
                // ... and stop - for now - where the linear index says the records behind it begin
                if (has_region) read_end = std::min(hard_read_end, (long long)(sp.reg_soft_end >> 16) + 0x10000 + 64);
                const long long seek_pos = ftell(fp);
                curp = Piece{};
                curp.n = fill(buf, 0);
                curp.eof = curp.n < last_want;
                curp.file_off = seek_pos;
                curp.first_off = (uint32_t)first_skip;
                first_skip = 0;
                if (!list_blocks(curp)) return false;
            } else {
                read_end = -1;
                partial_tail_ok = false;
                first_skip = 0;
                has_region = false;
            }
        } else {
            has_region = false;
        }
    }
    if (!submit_front(curp, n_submitted++) || !submit_back(curp)) return false;
    t_header = since(t_head0);

    // ---- the pieces ---------------------------------------------------------------------------------------------
    Contig cur;
    bool have_cur = false;
    int cur_ref = -1;
    std::set<int> seen;
    size_t n_pieces = 0, n_rows_total = 0, n_records = 0;
    auto finish_contig = [&]() -> bool {  // sort the finished contig's rows by fragment start and hand it out
        if (!have_cur) return true;
        clk.lap(3);
        DevColumns& U = *cur.dev;
        const size_t n = U.rows;
        bool ok = true;
        if (n) {
            std::shared_ptr<DevColumns> sorted(new DevColumns());
            sorted->device = device;
            sorted->bam = true;
            const size_t tmp_bytes = ftk::bam_sort_tmp_bytes(n);
            size_t tmp_cap = 0;
            void* tmp = device_cache().take(tmp_bytes, device, &tmp_cap, false);
            ok = tmp && sorted->reserve(n, pstream, true) &&
                 ftk::bam_sort_contig(pstream, n, U.start, U.end, U.mapq, U.strand, U.r1s, U.r1e, sorted->start, sorted->end,
                                      sorted->mapq, sorted->strand, sorted->r1s, sorted->r1e, sorted->ord, tmp, tmp_bytes) == 0 &&
                 hipGetLastError() == hipSuccess && hipStreamSynchronize(pstream) == hipSuccess;  // the unsorted block and the scratch go back
            if (tmp) device_cache().give(tmp, tmp_cap, device);
            if (!ok) {
                (void)hipGetLastError();
                return fail(FTK_ERR_OOM, "cannot sort the contig's rows on the device");
            }
            sorted->rows = n;
            if (hipEventCreateWithFlags(&U.ready, hipEventDisableTiming) == hipSuccess) (void)hipEventRecord(U.ready, pstream);
            cur.dev = sorted;
            emitted_refs.insert(cur_ref);
            if (!emit_device_bam(std::move(cur))) return false;
        }
        cur = Contig{};
        have_cur = false;
        cur_ref = -1;
        clk.lap(4);
        return true;
    };
    auto take_run = [&](int ref, DevSet& S, size_t r0, size_t r1) -> bool {
        if (r1 <= r0) return true;
        if (have_cur && ref != cur_ref && !finish_contig()) return false;
        if (!have_cur) {
            if (!seen.insert(ref).second)
                return fail(FTK_ERR_UNSORTED, ("contig " + ref_names[ref] + " appears in two separate runs: the BAM is not coordinate-sorted").c_str());
            cur_ref = ref;
            cur.name = ref_names[ref];
            cur.length = ref_lens[ref];
            cur.dev.reset(new DevColumns());
            cur.dev->device = device;
            cur.dev->bam = true;
            have_cur = true;
        }
        if (!cur.dev->append(S.d_s + r0, S.d_e + r0, S.d_q + r0, S.d_t + r0, r1 - r0, hipMemcpyDeviceToDevice, pstream, S.d_r1s + r0,
                             S.d_r1e + r0))
            return fail(FTK_ERR_OOM, "out of device memory for the contig's columns");
        return true;
    };
    auto settle = [&](Piece& pc) -> bool {
        DevSet& S = sets[pc.slot];
        if (hipEventSynchronize(S.done) != hipSuccess) {
            (void)hipGetLastError();
            return fail(FTK_ERR_HIP, "the device record parser failed");
        }
        clk.lap(1);
        S.pending = false;
        if (!pc.on_host && S.h_ist->n_bad) return fail(FTK_ERR_FORMAT, "BGZF inflate failed");
        for (size_t i = 0; i < S.n_tab; ++i)
            if (S.h_crc[i] != S.want_crc[i]) return fail(FTK_ERR_FORMAT, "BGZF block CRC mismatch (device inflate)");
        const ftk::BamSummary& B = *S.h_bsum;
        if (B.carry_overflow || !B.consistent || B.n_runs > (uint32_t)ftk::kBamMaxRuns || B.n_rows > S.max_lines) {
            if (clk.on)
                fprintf(stderr, "[ftk stream bam] piece %zu: the device could not settle the record chain (carry overflow %u, "
                                "consistent %u, runs %u, rows %u of %zu, range %u bytes in %u stretches, %u serial repairs, first "
                                "unsettled stretch %u: start %u, predecessor landed %u started %u, before it landed %u%s): the host "
                                "decoder takes over\n",
                        n_pieces, B.carry_overflow, B.consistent, B.n_runs, B.n_rows, S.max_lines, B.m, B.n_stretch, B.n_repairs,
                        B.first_unsettled, B.dbg[0], B.dbg[1], B.dbg[2], B.dbg[3], pc.on_host ? ", host-inflated" : "");
            want_host_restart = true;  // the host decoder takes the file (contigs handed out so far are skipped)
            return false;
        }
        if (B.bad) return fail(FTK_ERR_FORMAT, "corrupt BAM record");
        if (pc.eof && B.landing != B.m && !partial_tail_ok) return fail(FTK_ERR_FORMAT, "truncated BAM record");
        std::vector<std::pair<uint32_t, int>> runs(B.n_runs);
        for (uint32_t r = 0; r < B.n_runs; ++r) runs[r] = {B.run_row[r], B.run_ref[r]};
        std::sort(runs.begin(), runs.end());
        for (size_t r = 0; r < runs.size(); ++r) {
            const size_t r0 = runs[r].first, r1 = r + 1 < runs.size() ? runs[r + 1].first : (size_t)B.n_rows;
            if (!take_run(runs[r].second, S, r0, r1)) return false;
        }
        ++n_pieces;
        n_rows_total += B.n_rows;
        n_records += B.n_records;
        skipped[0] += B.n_unrepresentable;
        skipped[1] += B.n_nocigar_reverse;
        last_key = std::max(last_key, (unsigned long long)B.last_key);
        S.freed_valid = hipEventRecord(S.freed, pstream) == hipSuccess;
        if (!S.freed_valid) {
            (void)hipGetLastError();
            return fail(FTK_ERR_HIP, "cannot record a buffer set's release");
        }
        clk.lap(2);
        return true;
    };

    for (;;) {
        // The backs go onto the parse stream in file order: curp's now (waiting for the host threads if they hold
        // it), the following ones as far as they are ready without waiting.  BEFORE any new front: the piece read
        // next goes into the slot of the piece settled last, whose text the parse of ITS successor (curp) still
        // reads the carry from - submit_back re-records that slot's `freed` behind curp's parse.
        if (!curp.back_done && !submit_back(curp)) return false;
        for (auto& pc : ahead) {
            if (pc.back_done) continue;
            if (pc.on_host && host_job[pc.slot].valid() &&
                host_job[pc.slot].wait_for(std::chrono::seconds(0)) != std::future_status::ready)
                break;
            if (!submit_back(pc)) return false;
        }
        // read and enqueue the next pieces before this one is settled
        if (!read_ahead_fronts()) return false;
        for (auto& pc : ahead) {  // (the fronts just enqueued: GPU pieces' backs follow at once)
            if (pc.back_done) continue;
            if (pc.on_host && host_job[pc.slot].valid() &&
                host_job[pc.slot].wait_for(std::chrono::seconds(0)) != std::future_status::ready)
                break;
            if (!submit_back(pc)) return false;
        }
        clk.lap(5);
        if (!settle(curp)) return false;
        if (curp.eof) {
            // (region reads) the read stopped at the linear index's hint: complete when the last record lies at or behind
            // the region's end or on a later reference; else a record longer than an index window hid the true end
            const bool soft = has_region && read_end >= 0 && read_end < hard_read_end && curp.file_off >= 0 &&
                              curp.file_off + (long long)curp.n >= read_end;
            if (!soft) break;
            const long long lref = (long long)(uint32_t)(last_key >> 32), lpos = (long long)(uint32_t)last_key;
            const bool complete = last_key != 0 && (lref != (long long)region_ref || lpos >= reg_stop);
            if (clk.on)
                fprintf(stderr, "[ftk stream bam] region %lld-%lld: piece ends at file offset %lld (hint %lld, contig ends %lld): %s\n",
                        reg_start, reg_stop, curp.file_off + (long long)curp.n, read_end, hard_read_end,
                        complete ? "complete" : "a long record hides the end, reading on");
            if (complete) break;
            read_end = std::min(hard_read_end, read_end + (long long)(size_t(8) << 20));
            curp.eof = false;
            if (!read_ahead_fronts()) return false;
        }
        if (ahead.empty()) return fail(FTK_ERR_HIP, "piece queue out of step");
        curp = std::move(ahead.front());
        ahead.pop_front();
        {
            std::lock_guard<std::mutex> lk(mu);
            if (stop) return false;
        }
    }
    if (!finish_contig()) return false;
    clk.lap(4);
    clk.report("bam, records parsed on the device (inflate = waiting for a piece, parse = appends, merge = sort)");
    if (clk.on)
        fprintf(stderr, "[ftk stream bam] %zu pieces, %zu records, %zu fragments parsed on the device (stretch %u bytes); of \"other\": "
                        "header piece %.1f ms, fronts enqueued %.1f ms, waiting for the host threads' inflate %.1f ms\n",
                n_pieces, n_records, n_rows_total, stretch_bytes, t_header, t_front, t_jobwait);
    return true;
}

extern "C" {

// Give back what the library keeps for reuse between calls: idle page-locked blocks (decoded tables, result arrays),
// idle device blocks of parsed contigs, the streams' idle buffer sets.  Nothing in use is touched.
int64_t ftk_cache_trim(void) {
    size_t n = table_cache().trim() + result_cache().trim() + plain_result_cache().trim();
    if (have_hip_device()) {
        n += device_cache().trim() + devset_pool().trim();
        (void)stream_pool().trim();
        n += ftk::inflate_release_scratch();
    }
    return (int64_t)n;
}

int ftk_fragfile_index_contigs(const char* path, char* names_out, int64_t cap, int64_t* needed_out, int* is_bed6_out) {
    if (!path || !needed_out) return dfail(FTK_ERR_INVALID, "NULL argument");
    *needed_out = 0;
    Bytes raw, img;
    const std::string ipath = std::string(path) + ".tbi";
    if (!read_file(ipath.c_str(), &raw) || raw.size() < 8) return dfail(FTK_ERR_FORMAT, "no usable tabix index at %s", ipath.c_str());
    if (inflate_all(raw, 1, &img) != FTK_OK || img.size() < 36 || memcmp(img.data(), "TBI\1", 4) != 0)
        return dfail(FTK_ERR_FORMAT, "%s is not a tabix index", ipath.c_str());
    const uint8_t* p = img.data();
    const int32_t n_ref = rd_i32(p + 4), l_nm = rd_i32(p + 32);
    if (n_ref < 0 || l_nm < 0 || 36 + (size_t)l_nm > img.size()) return dfail(FTK_ERR_FORMAT, "corrupt tabix index");
    // names of the references that hold at least one chunk, newline-separated
    std::string joined;
    size_t a = 36;
    for (int k = 0; k < n_ref && a < 36 + (size_t)l_nm; ++k) {
        const char* nm = (const char*)p + a;
        const size_t len = strnlen(nm, 36 + (size_t)l_nm - a);
        const IndexSpan sp = index_lookup(ipath, false, std::string(nm, len), -1);
        if (!sp.usable) return dfail(FTK_ERR_FORMAT, "corrupt tabix index");
        if (sp.present) { joined.append(nm, len); joined.push_back('\n'); }
        a += len + 1;
    }
    *needed_out = (int64_t)joined.size() + 1;
    if (names_out && cap >= (int64_t)joined.size() + 1) memcpy(names_out, joined.c_str(), joined.size() + 1);
    if (is_bed6_out) {  // layout of the first data row (io/alignment.py:143-156), from the file's first block
        *is_bed6_out = 0;
        FILE* fp = fopen(path, "rb");
        if (!fp) return dfail(FTK_ERR_IO, "cannot read %s", path);
        Bytes head, text;
        head.alloc(1 << 17);
        const size_t got = fread(head.data(), 1, head.size(), fp);
        fclose(fp);
        size_t bs = 0;
        const size_t q = gzip_header(head.data(), got, 0, &bs);
        if (q && bs && bs <= got) {
            Bytes one;
            one.alloc(bs);
            memcpy(one.data(), head.data(), bs);
            if (inflate_all(one, 1, &text) == FTK_OK) {
                const char* b = (const char*)text.data();
                const char* e = b + text.size();
                while (b < e) {
                    const char* nl = (const char*)memchr(b, '\n', (size_t)(e - b));
                    const char* le = nl ? nl : e;
                    if (le > b && *b != '#') {
                        int tabs = 0;
                        for (const char* x = b; x < le; ++x) tabs += (*x == '\t');
                        *is_bed6_out = (tabs + 1) > 5;
                        break;
                    }
                    if (!nl) break;
                    b = nl + 1;
                }
            }
        }
    }
    return FTK_OK;
}

static int fragstream_open_impl(const char* path, const char* contig, int is_bam, int n_threads, int max_queued,
                                int device, ftk_fragstream** out, long long reg_start = -1, long long reg_stop = -1) {
    if (!path || !out) return dfail(FTK_ERR_INVALID, "NULL argument");
    *out = nullptr;
    FILE* fp = fopen(path, "rb");
    if (!fp) return dfail(FTK_ERR_IO, "cannot read %s", path);
    ftk_fragstream* s = new ftk_fragstream();
    s->path = path;
    if (contig) { s->only = contig; s->has_only = true; }
    if (contig && reg_start >= 0 && reg_stop > reg_start) {
        s->has_region = true;
        s->reg_start = reg_start;
        s->reg_stop = reg_stop;
    }
    s->bam = is_bam != 0;
    s->n_threads = std::max(1, n_threads);
    s->max_queued = (size_t)std::max(1, max_queued);
    s->fp = fp;
    // BAM records are parsed by the host; FTK_DEVICE_PARSE=0 keeps text rows there too
    static const bool device_parse = !(getenv("FTK_DEVICE_PARSE") && atoi(getenv("FTK_DEVICE_PARSE")) == 0);
    s->device = (!s->bam && device_parse) ? device : -1;
    s->inflate_device = s->bam ? device : -1;  // BAM: records stay on the host, the BGZF inflate may use the GPU
    s->producer = std::thread([s] { s->run(); });
    *out = s;
    return FTK_OK;
}

int ftk_fragstream_open(const char* path, const char* contig, int is_bam, int n_threads, int max_queued,
                        ftk_fragstream** out) {
    return fragstream_open_impl(path, contig, is_bam, n_threads, max_queued, -1, out);
}

int ftk_fragstream_open_device(int device_id, const char* path, const char* contig, int is_bam, int n_threads,
                               int max_queued, ftk_fragstream** out) {
    if (device_id < 0 || !have_hip_device()) return dfail(FTK_ERR_NO_DEVICE, "ftk_fragstream_open_device: no HIP device");
    return fragstream_open_impl(path, contig, is_bam, n_threads, max_queued, device_id, out);
}

int ftk_fragstream_open_region(int device_id, const char* path, const char* contig, int64_t start, int64_t stop, int is_bam,
                               int n_threads, int max_queued, ftk_fragstream** out) {
    if (!contig || start < 0 || stop <= start) return dfail(FTK_ERR_INVALID, "a region needs a contig and 0 <= start < stop");
    return fragstream_open_impl(path, contig, is_bam, n_threads, max_queued, device_id, out, (long long)start, (long long)stop);
}

int ftk_fragstream_next(ftk_fragstream* s, ftk_fragtable** out) {
    if (!s || !out) return dfail(FTK_ERR_INVALID, "NULL argument");
    *out = nullptr;
    std::unique_lock<std::mutex> lk(s->mu);
    s->consumer_waiting = true;
    s->cv.wait(lk, [&] { return !s->ready.empty() || s->finished; });
    s->consumer_waiting = false;
    if (!s->ready.empty()) {
        *out = s->ready.front();
        s->ready.pop_front();
        s->cv.notify_all();
        return FTK_OK;
    }
    if (s->err != FTK_OK) return dfail(s->err, "%s", s->errmsg.c_str());
    return FTK_OK;  // end of file: *out stays NULL
}

int ftk_fragstream_n_refs(ftk_fragstream* s) {
    if (!s) return 0;
    std::unique_lock<std::mutex> lk(s->mu);
    s->cv.wait(lk, [&] { return s->header_ready || s->finished; });
    return (int)s->ref_names.size();
}
const char* ftk_fragstream_ref_name(ftk_fragstream* s, int i) {
    if (!s || i < 0 || i >= ftk_fragstream_n_refs(s)) return nullptr;
    return s->ref_names[i].c_str();
}
int64_t ftk_fragstream_ref_length(ftk_fragstream* s, int i) {
    if (!s || i < 0 || i >= ftk_fragstream_n_refs(s)) return -1;
    return s->ref_lens[i];
}

int ftk_fragstream_skipped(ftk_fragstream* s, int64_t out[2]) {
    if (!s || !out) return FTK_ERR_INVALID;
    out[0] = s->skipped[0].load();
    out[1] = s->skipped[1].load();
    return FTK_OK;
}

int ftk_fragtable_skipped(const ftk_fragtable* t, int64_t out[2]) {
    if (!t || !out) return FTK_ERR_INVALID;
    out[0] = t->skipped[0];
    out[1] = t->skipped[1];
    return FTK_OK;
}

int ftk_fragstream_stage_ms(ftk_fragstream* s, double out[6]) {
    if (!s || !out) return FTK_ERR_INVALID;
    std::lock_guard<std::mutex> lk(s->mu);
    for (int k = 0; k < 6; ++k) out[k] = s->stage_ms[k];
    return FTK_OK;
}

void ftk_fragstream_close(ftk_fragstream* s) {
    if (!s) return;
    {
        std::lock_guard<std::mutex> lk(s->mu);
        s->stop = true;
        s->cv.notify_all();
    }
    if (s->producer.joinable()) s->producer.join();
    s->drain_ahead();
    for (auto* t : s->ready) delete t;
    if (s->pstream) {
        (void)hipSetDevice(s->device >= 0 ? s->device : s->inflate_device);
        (void)hipStreamSynchronize(s->pstream);
        stream_pool().give(s->device >= 0 ? s->device : s->inflate_device, s->pstream);
    }
    if (s->fp) fclose(s->fp);
    delete s;
}

}  // extern "C"
