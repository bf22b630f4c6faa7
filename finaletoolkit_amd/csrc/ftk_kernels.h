// Launchers of the HIP kernels in ftk_kernels.hip (internal).
#pragma once

#include "ftk_internal.h"

namespace ftk {

// per-wave LDS histograms are used up to this many bins (4 waves x bins x 4 B)
constexpr int kHistSmallMaxBins = 2048;
// block-wide LDS histogram limit (128 KiB of the CU's 160 KiB)
constexpr int kHistMaxBins = 32768;

struct FragStats {
    int unsorted;
    int max_len;
    int min_len;
    int max_end;
    int min_start;
};

// Device scratch describing where each window's candidate fragments are.
struct WindowPlan {
    int32_t* cand_lo;     // [n_win]
    int32_t* cand_hi;     // [n_win]
    uint32_t* nchunks;    // [n_win] 0 = wave-per-window path
    uint32_t* chunk_off;  // [n_win + 1]
};

struct WpsParams {
    long long start, stop;  // single-interval form
    long long chrom_size;
    int hl, hr, odd;
    int min_len, max_len, mapq_min;
    int lmax;  // min(max_len, longest fragment of the contig)
    int nt_store;  // non-temporal score stores (default on)
    int xcd_remap; // XCD-contiguous block -> tile map (experiment, default off: measured slower)
};

struct CleaveParams {
    long long start, stop;  // single-interval form
    int min_len, max_len, mapq_min;
    int lmax;  // longest admissible fragment
};

void launch_stats(hipStream_t s, const int32_t* start, const int32_t* end, int n, FragStats* st);
void launch_bin_index(hipStream_t s, const int32_t* start, int n, int n_bins, int32_t* idx);
// *bad |= 1 when some fragment does not hold its read1 span (ContigView::r1_inside)
void launch_r1_inside(hipStream_t s, const int32_t* start, const int32_t* end, const int32_t* r1s, const int32_t* r1e,
                      int n, int* bad);
void launch_plan(hipStream_t s, const ContigView& cv, const int32_t* ws, const int32_t* we, int n_win, int lmax,
                 int small_max, const WindowPlan& pl, int64_t* const zero[4]);

// Where the motif pass reads its k-mers (device pointers) and which ends count.
struct MotifParams {
    const uint8_t* img;        // reference image of the contig (ftk_ref_upload)
    const int32_t* nblk_start; // 2bit: N blocks, sorted
    const int32_t* nblk_end;
    int n_nblk;
    int kind;                  // FTK_REF_*
    int chrom_len;
    int line_bases, line_width;  // FASTA text geometry
    int k, f_off, r_off;
    int both, neg, guard, rev_err;
};

// What one window-feature pass should produce (NULL output = feature off).
struct FeatureRequest {
    const ftk_filter* filter = nullptr;  // coverage + histogram predicate
    int64_t* cov_out = nullptr;
    uint32_t* hist_out = nullptr;        // [n_win][n_bins]; zero-filled by the caller only when small_path is off
    int64_t* over_out = nullptr;         // (same)
    int len_lo = 0, n_bins = 0;
    int64_t* short_out = nullptr;        // DELFI
    int64_t* long_out = nullptr;
    int delfi_mapq_min = 0;
    ftk_gaps gaps{};
    const int32_t* bl_off = nullptr;
    const int32_t* bl_r0 = nullptr;
    const int32_t* bl_pm = nullptr;
    // motif histogram instead of the length histogram (hist_out = [n_win][4^k], over_out = errors)
    const MotifParams* motif = nullptr;
    int block_threads = 256;  // block-per-window path: 512 for windows of several thousand candidates
};
// One contig's share of a batched window-feature launch (device pointers).
struct FeatItem {
    ContigView cv;
    const int32_t* ws;
    const int32_t* we;
    const int32_t* bl_off;  // per-window blacklist CSR of this item (NULL: none)
    const int32_t* bl_r0;
    const int32_t* bl_pm;
    int32_t win_base;       // first row of this item in the concatenated outputs
    int32_t n_win;
    int32_t lmax;
    int32_t cen0, cen1, tel0, tel1;  // gap constants, clamped (gap_constants)
};
void gap_constants(const ftk_gaps& g, int out[4]);
void launch_window_features_batch(hipStream_t s, const FeatItem* d_items, int n_items, int total_win,
                                  const FeatureRequest& r, bool bam);
// block_lmax >= 0: one block per window (feat_block_kernel, needs no plan and no zeroed outputs), with
// block_lmax = longest fragment any requested feature can accept; < 0: the planned small + chunked passes.
// tail: a whole-interval WPS of the same contig to run in the SAME launch behind the feature blocks (FAST block
// path only); returns true when it was merged, false when the caller has to launch it itself.
struct WpsTail {
    WpsParams p;
    int64_t n_tiles;
    int64_t* out;
};
bool launch_window_features(hipStream_t s, int grid_large, const ContigView& cv, const int32_t* ws, const int32_t* we,
                            int n_win, const WindowPlan& pl, const FeatureRequest& r, bool small_path,
                            int block_lmax = -1, const WpsTail* tail = nullptr);
void launch_add_i64(hipStream_t s, const int64_t* a, const int64_t* b, int64_t* out, int n);
// per-window length statistics from dense histogram rows: out[w][7] = mean median stdev min max total n_short
void launch_window_stats(hipStream_t s, const uint32_t* hist, int n_win, int n_bins, int len_lo, int short_cut, double* out);
void launch_wps(hipStream_t s, const ContigView& cv, const WpsParams& p, int64_t n_tiles, const int64_t* iv_start,
                const int64_t* iv_stop, const int64_t* out_off, const int32_t* tile_iv, const int32_t* tile_k,
                int64_t* out);
// Window features computed inside the WPS pass (regular bin tiling, midpoint policy).
struct FusedParams {
    int win_start, win_len, n_win;
    int bam;  // read1 fetch semantics (the contig has read1 columns)
    int ch_q, ch_min, ch_max;
    int do_cov, do_hist, len_lo, n_bins;
    int do_delfi, df_q, cen0, cen1, tel0, tel1;
    const int32_t* bl_off;
    const int32_t* bl_r0;
    const int32_t* bl_pm;
    int64_t* cov_out;
    uint32_t* hist_out;
    int64_t* over_out;
    int64_t* short_out;
    int64_t* long_out;
};
void launch_wps_fused(hipStream_t s, const ContigView& cv, const WpsParams& p, int64_t n_tiles, const FusedParams& F,
                      int64_t* out);
// One (contig, interval) of a batched WPS launch.
struct WpsItem {
    ContigView cv;
    long long start, stop, chrom_size;
    long long out_off;    // index of the interval's first score in the output
    long long tile_base;  // first tile of this item in the launch
    int lmax;
    int pad_;
};
void launch_wps_batch(hipStream_t s, const WpsParams& p, const WpsItem* d_items, int n_items, int64_t n_tiles,
                      int64_t* out);
void launch_cleavage(hipStream_t s, const ContigView& cv, const CleaveParams& p, int64_t n_tiles,
                     const int64_t* iv_start, const int64_t* iv_stop, const int64_t* out_off, const int32_t* tile_iv,
                     const int32_t* tile_k, double* out);
// One block's share of an adjust_wps run: n_out outputs starting at output o0 of
// a run with m outputs; inputs start at scores[in_base], results at out[out_base].
struct AdjustTile {
    int64_t in_base;
    int64_t out_base;
    int32_t n_out;
    int32_t o0;
    int32_t m;
    int32_t interval;
};
constexpr int kAdjustMaxWindow = 2048;
constexpr int kAdjustFastTile = 4096;  // outputs per tile of the histogram median (64 lanes x 64 consecutive outputs)
int adjust_sort_size(int W, int* tile_out);
// fast_tiles / todo (may be NULL): the tiles of the histogram median (at most kAdjustFastTile outputs each) and n_iv + 2
// ints (cleared by launch_adjust_tiles) in which it marks the intervals it leaves to the sort kernel
// The tile lists of both kinds from the run offsets, on the device, in two launches.  Counts: pre0 / pre1 (n_iv + 1 ints
// each) = tiles of the sort kernel (tile0 outputs each) / of the histogram median (tile1; 0: none) in front of every run;
// offs_src: the offsets in device-readable memory (page-locked host memory will do), copied to offs (n_iv + 1 slots in
// HBM, what the other kernels read); todo (may be NULL): the histogram kernels' marks, n_iv + 2 ints, cleared.  The
// launch touches nothing but these arrays whatever the offsets hold - it may go out before they are validated.
void launch_adjust_tile_counts(hipStream_t s, const int64_t* offs_src, int64_t* offs, int n_iv, int W, int tile0, int tile1,
                               int* pre0, int* pre1, int* todo);
// Fill: t0[n0] / t1[n1] (t1 NULL: none) from the counts.
void launch_adjust_tile_fill(hipStream_t s, const int64_t* offs, int n_iv, int W, int tile0, int tile1, const int* pre0,
                             const int* pre1, AdjustTile* t0, int n0, AdjustTile* t1, int n1);
void launch_adjust_filter(hipStream_t s, const double* scores, const AdjustTile* tiles, int n_tiles,
                          const double* edge_sub, int W, int use_mean, double* out, const AdjustTile* fast_tiles = nullptr,
                          int n_fast_tiles = 0, int* todo = nullptr, int n_iv = 0);
void launch_savgol(hipStream_t s, const double* adj, const AdjustTile* tiles, int n_tiles, const double* coef,
                   const double* edge, int sw, double* out);
void launch_gc_count(hipStream_t s, const uint8_t* img, int64_t img_bytes, int kind, const int64_t* lo,
                     const int64_t* hi, int n, int64_t* out);
void launch_select_count(hipStream_t s, const ContigView& cv, int lo, int hi, int ws, int we, const ftk_filter& f,
                         uint32_t* block_cnt);
void launch_scan_u32(hipStream_t s, const uint32_t* in, int n, uint32_t* off);
void launch_select_write(hipStream_t s, const ContigView& cv, int lo, int hi, int ws, int we, const ftk_filter& f,
                         const uint32_t* block_off, int64_t cap, int32_t* len_out, int32_t* start_out,
                         int32_t* end_out, uint8_t* mapq_out, uint8_t* strand_out, int32_t* order_out = nullptr);

}  // namespace ftk
