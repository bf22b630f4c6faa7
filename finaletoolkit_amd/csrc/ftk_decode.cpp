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

#include "ftk_stream_core.inc"  // the streaming decoder
#include "ftk_stream_text.inc"  // text streams
#include "ftk_stream_bam.inc"  // BAM streams
#include "ftk_stream_api.inc"  // the C entry points of the streams and the caches (`ftk_fragstream_*`, `ftk_cache_trim`, `ftk_fragfile_index_contigs`)
