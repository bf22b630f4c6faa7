// Internal declarations shared by the HIP translation units of libftk_hip.so.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <string>
#include <vector>

#include "ftk.h"

namespace ftk {

// Coarse position index: bin_idx[k] = first fragment whose start >= k << kBinShift.
constexpr int kBinShift = 9;
// Work-item size (fragments) of the chunked large-window path.
constexpr int kChunk = 4096;
// Candidate ranges up to this many fragments are handled one-wave-per-window.
constexpr int kSmallMax = 1024;
// start / end of the padding fragments behind every contig's columns (coordinates are < 2^30).
constexpr int kPadCoord = 1 << 30;
// Positions per WPS tile.
constexpr int kWpsTile = 4096;

// One contig's fragments, resident in HBM (start-sorted SoA, 10 B / fragment).
struct ContigView {
    const int32_t* start;
    const int32_t* end;
    const uint8_t* mapq;
    const uint8_t* strand;
    const int32_t* r1_start;  // BAM only (else nullptr)
    const int32_t* r1_end;
    const int32_t* order;     // BAM, optional: file-order rank (else nullptr)
    const int32_t* bin_idx;   // n_bins + 1 entries
    int32_t n;                // fragments
    int32_t n_bins;
    int32_t max_len;          // longest fragment in the contig
    // BAM: 1 when every fragment holds its read1 span (start <= r1_start < r1_end <= end; checked on the device by
    // ftk_frags_set_read1).  A fragment that lies inside a window then has its read1 inside it too, so the read1
    // fetch test (io/alignment.py:245) can only fail for fragments that cross a window bound and the kernels read
    // the read1 columns for those alone.  0: the columns are read for every fragment.
    int32_t r1_inside;
};

struct ContigData {
    ContigView v{};
    void* base = nullptr;     // one allocation holding all columns
    int32_t* r1 = nullptr;    // optional allocation for read1 columns
    int32_t* order = nullptr; // optional allocation for the file-order column
    int32_t* bin_idx = nullptr;
    int64_t n = 0;
    int32_t max_len = 0;
    int32_t max_end = 0;
};

// Device-resident DELFI metadata of one (contig, bins, blacklist) combination:
// the windows and the per-window blacklist CSR.  Built once, reused while the
// caller keeps passing the same arrays (the reference parses its blacklist
// once per worker too, frag/_delfi.py:65-107).
struct DelfiMeta {
    uint64_t key = 0, key2 = 0;
    int contig_id = -1;
    int64_t n_win = 0, n_bl = 0;
    size_t n_r = 0;
    void* base = nullptr;  // one allocation
    int32_t *d_ws = nullptr, *d_we = nullptr, *d_off = nullptr, *d_r0 = nullptr, *d_pm = nullptr;
};

}  // namespace ftk

struct ftk_ctx {
    int device = -1;
    int n_cu = 256;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    std::vector<hipEvent_t> ev_slots;  // lazily created, FTK_MAX_EVENTS entries
    std::map<int, ftk::ContigData> contigs;
    std::map<int, std::string> names;  // contigs loaded by ftk_frags_load_*
    std::string err;
    std::vector<ftk::DelfiMeta> delfi_cache;
    struct RefImage {
        void* d = nullptr;
        int64_t bytes = 0;
        int64_t cap = 0;  // size of the device block (>= bytes + 32; blocks are recycled through ref_pool)
        int kind = 0;
        // ftk_ref_set_layout (needed by the motif pass)
        int64_t chrom_len = -1;
        int32_t line_bases = 0, line_width = 0;
        int32_t* d_nblk = nullptr;  // [2][n_nblk]: starts then ends
        int32_t n_nblk = 0;
    };
    std::map<int, RefImage> refs;  // reference-sequence images for the DELFI GC count
    // released images' device blocks, kept for the next upload (hipMalloc / hipFree wait for the whole device - the
    // decoder's kernels included - and a genome-wide DELFI run uploads one image per contig), and the two
    // page-locked staging chunks ftk_ref_upload_file reads the file through
    std::vector<std::pair<void*, size_t>> ref_pool;
    void* ref_stage[2] = {nullptr, nullptr};
    hipEvent_t ref_stage_done[2] = {nullptr, nullptr};
    // per-base results on a narrow wire (ftk_wps with a host output): two page-locked chunks of 16-bit scores and the
    // events behind their copies
    void* narrow_stage[2] = {nullptr, nullptr};
    hipEvent_t narrow_done[2] = {nullptr, nullptr};
    // batched launches: the per-item descriptors last uploaded (re-used while the caller repeats the batch)
    std::vector<unsigned char> batch_host[2];  // [0] window features, [1] WPS
    void* batch_dev[2] = {nullptr, nullptr};
    size_t batch_cap[2] = {0, 0};
    // small host arrays on their way to the device by DMA (a pageable copy above 16 KB takes the runtime's slow route:
    // ~28 us for 80 KB); page-locked, grow-only, used by calls that synchronise before they return
    void* param_stage = nullptr;
    size_t param_stage_bytes = 0;
    // grow-only device scratch, reused by every call (stream-ordered)
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    void* d_stats = nullptr;  // load-time validation summary (upload_common)
    // asynchronous host results (ftk_wps_async): two device buffers, each copied back on the copy stream behind
    // its kernel; a buffer is reused once its copy has finished
    hipStream_t copy_stream = nullptr;
    void* abuf[2] = {nullptr, nullptr};
    size_t abuf_bytes[2] = {0, 0};
    hipEvent_t a_kernel_done[2] = {nullptr, nullptr}, a_copy_done[2] = {nullptr, nullptr};
    bool a_pending[2] = {false, false};
    int a_next = 0;
    int a_token[2] = {-1, -1};  // the token whose result sits in each buffer; tokens count up, token & 1 is its buffer
    int a_issued = 0;           // tokens handed out so far
};
