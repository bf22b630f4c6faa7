/*
 * ftk.h -- C ABI of the MI355X fragment-feature engine (libftk_hip.so).
 *
 * This is the drop-in boundary for FinaleToolkit's per-window hot path.  The
 * reference (epifluidlab/FinaleToolkit 1.1.0) is pure Python and has no FFI of
 * its own; every entry point below names the reference loop it replaces
 * (paths relative to the reference checkout, `src/finaletoolkit/...`).
 *
 * Conventions
 *  - extern "C", plain pointers and sizes, no C++/torch types.
 *  - Every call returns an int: FTK_OK (0) or a negative FTK_ERR_* code.
 *    ftk_last_error(ctx) returns a message owned by the library, valid until
 *    the next call on that ctx (ctx may be NULL for context-less calls).
 *  - One ftk_ctx per GPU.  A ctx is single-threaded (caller serialises);
 *    different ctxs may be driven from different threads.
 *  - Coordinates are 0-based half-open int32 (same as the reference's
 *    Fragment record, io/alignment.py:25-54).  Fragments of one contig are
 *    held as a start-sorted SoA (start i32, end i32, mapq u8, strand u8) in
 *    HBM: 10 bytes per fragment.
 *  - "in" pointers for windows/intervals and "out" pointers for results may be
 *    host OR device pointers; the library detects which and copies only when
 *    it has to.  Results are complete when the call returns for host
 *    pointers; for device pointers they are ordered on the ctx stream
 *    (ftk_ctx_sync / ftk_timer_stop to wait).
 *  - There is NO CPU fallback inside this library: without a usable gfx950
 *    device ftk_ctx_create fails with FTK_ERR_NO_DEVICE.
 */
#ifndef FTK_H
#define FTK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FTK_OK 0
#define FTK_ERR_INVALID (-1)   /* bad argument / out-of-domain input       */
#define FTK_ERR_NO_DEVICE (-2) /* no usable HIP device                      */
#define FTK_ERR_HIP (-3)       /* a HIP runtime call failed                 */
#define FTK_ERR_OOM (-4)       /* host or device allocation failed          */
#define FTK_ERR_IO (-5)        /* file could not be opened / read           */
#define FTK_ERR_FORMAT (-6)    /* file is not BGZF/gzip fragment text or BAM */
#define FTK_ERR_NO_CONTIG (-7) /* contig id / name not loaded               */
#define FTK_ERR_UNSORTED (-8)  /* fragments not sorted by start             */

/* Open window bounds (the reference's `start=None` / `stop=None`,
 * utils/_frag_generator.py:35-50). */
#define FTK_OPEN_LO INT32_MIN
#define FTK_OPEN_HI INT32_MAX
/* Open length bound (`min_length=None` / `max_length=None`,
 * utils/_comparison.py:13-24). */
#define FTK_LEN_OPEN (-1)

#define FTK_POLICY_MIDPOINT 0 /* utils/_frag_generator.py:35-42 */
#define FTK_POLICY_ANY 1      /* utils/_frag_generator.py:44-50 */
#define FTK_POLICY_FETCH 2    /* no intersect test: whatever the index query returns (AlignmentWrapper.fetch,
                               * io/alignment.py:216-240); taken by ftk_frag_select / ftk_frag_lengths only */

#define FTK_FETCH_TABIX 0     /* io/alignment.py:270-302: rows overlapping the window */
#define FTK_FETCH_BAM_READ1 1 /* io/alignment.py:242-268: read1 alignments overlapping the window */

#define FTK_MAX_TELOMERES 8

typedef struct ftk_ctx ftk_ctx;
typedef struct ftk_fragtable ftk_fragtable;

/* The shared per-fragment predicate of utils/_frag_generator.py:117-130 plus
 * the mapq cut of io/alignment.py:291 / :60-71. */
typedef struct ftk_filter {
    int32_t mapq_min;   /* keep mapq >= mapq_min                               */
    int32_t min_len;    /* keep len >= min_len; FTK_LEN_OPEN = no bound        */
    int32_t max_len;    /* keep len <= max_len; FTK_LEN_OPEN = no bound        */
    int32_t policy;     /* FTK_POLICY_*                                        */
    int32_t fetch_mode; /* FTK_FETCH_*                                         */
} ftk_filter;

/* Per-contig centromere/telomere constants of genome/gaps.py:202-267
 * (ContigGaps).  has_gaps == 0 means `contig_gaps is None`
 * (frag/_delfi.py:416-428,464). */
typedef struct ftk_gaps {
    int32_t has_gaps;
    int32_t cen_start, cen_stop;
    int32_t n_telo;
    int32_t telo_start[FTK_MAX_TELOMERES];
    int32_t telo_stop[FTK_MAX_TELOMERES];
} ftk_gaps;

/* ---- library / device ---------------------------------------------------- */
const char* ftk_version(void);
int ftk_device_count(int* n_out);
int ftk_ctx_create(int device_id, ftk_ctx** out);
void ftk_ctx_destroy(ftk_ctx* ctx);
const char* ftk_last_error(ftk_ctx* ctx);
/* Launch on a caller-provided hipStream_t (e.g. torch's current stream);
 * NULL restores the ctx's own stream. */
int ftk_ctx_set_stream(ftk_ctx* ctx, void* hip_stream);
int ftk_ctx_sync(ftk_ctx* ctx);
/* HIP-event stopwatch on the ctx stream (used by bench.py for per-kernel
 * durations).  ftk_timer_stop waits for the stop event. */
int ftk_timer_start(ftk_ctx* ctx);
int ftk_timer_stop(ftk_ctx* ctx, float* ms_out);
/* Event slots for timing individual launches without draining the stream:
 * record any number of slots inside a timed region, read the differences
 * afterwards (ftk_event_elapsed_ms waits for slot_b). */
#define FTK_MAX_EVENTS 4096
int ftk_event_record(ftk_ctx* ctx, int slot);
int ftk_event_elapsed_ms(ftk_ctx* ctx, int slot_a, int slot_b, float* ms_out);

/* ---- fragments: SoA residency in HBM -------------------------------------
 * Replaces the per-window `AlignmentWrapper.fetch` re-open + Fragment tuple
 * stream (io/alignment.py:217-302, utils/_frag_generator.py:112-116): one
 * contig is decoded once and stays resident. */
int ftk_frags_from_host(ftk_ctx* ctx, int contig_id, const int32_t* start, const int32_t* end,
                        const uint8_t* mapq, const uint8_t* strand, int64_t n);
/* Same, sources already in device memory (copied device-to-device into the
 * library's padded SoA). */
int ftk_frags_from_device(ftk_ctx* ctx, int contig_id, const int32_t* d_start, const int32_t* d_end,
                          const uint8_t* d_mapq, const uint8_t* d_strand, int64_t n);
/* BAM only: read1 alignment span per fragment, needed for FTK_FETCH_BAM_READ1
 * (io/alignment.py:245: pysam fetch returns read1 alignments overlapping the
 * window).  Arrays are parallel to the fragment arrays. */
int ftk_frags_set_read1(ftk_ctx* ctx, int contig_id, const int32_t* r1_start, const int32_t* r1_end, int64_t n);
/* BAM only, optional: file-order rank per fragment (ftk_fragtable_order).  With it ftk_frag_lengths /
 * ftk_frag_select return their rows in file order (pysam's iteration order) instead of start order. */
int ftk_frags_set_order(ftk_ctx* ctx, int contig_id, const int32_t* order, int64_t n);
int ftk_frags_info(ftk_ctx* ctx, int contig_id, int64_t* n_out, int32_t* max_len_out, int32_t* max_end_out);
int ftk_frags_release(ftk_ctx* ctx, int contig_id);

/* ---- fragment-file decoder (host only; no ctx / GPU needed) ---------------
 * io/alignment.py:270-302 (_fetch_tabix): BGZF/gzip text rows -> fragments.
 * Column layout: FinaleDB `chrom start stop mapq strand`; BED6 when the first
 * data row has > 5 columns (io/alignment.py:143-156) -> mapq = col 4, strand =
 * col 5.  Rows that do not parse are skipped (io/alignment.py:301-302).  No
 * mapq filtering here: mapq is a column, the cut is applied by the kernels.
 * io/alignment.py:242-268 (_fetch_sam): BAM records -> fragments (flag filter,
 * read1 only, TLEN reconstruction: TLEN > 0 -> [pos, pos + TLEN), CIGAR or not; TLEN < 0 ->
 * [end + TLEN, end) with end = htslib's bam_endpos = pos + max(reference length of the CIGAR, 1));
 * the read1 span kept for region queries is [pos, bam_endpos); rows sorted by fragment start. */
int ftk_fragfile_decode(const char* path, const char* contig /* NULL = all */, int n_threads, ftk_fragtable** out);
int ftk_bam_decode(const char* path, const char* contig /* NULL = all */, int n_threads, ftk_fragtable** out);
/* The same decoders as a stream: a producer thread decodes the file piece by piece (BGZF blocks in
 * parallel on n_threads) and hands over every finished contig as a one-contig table in page-locked
 * memory, at most max_queued contigs ahead of the caller -- so the caller uploads and computes on
 * contig k while contig k+1 is being decoded, and host memory stays bounded for any file size.
 * ftk_fragstream_next returns FTK_OK with *out == NULL at the end of the file; tables are freed with
 * ftk_fragtable_free.  Input must keep each contig's rows together (coordinate-sorted), else
 * FTK_ERR_UNSORTED.  BAM: ftk_fragstream_n_refs / _ref_name / _ref_length give the header's @SQ list
 * (they wait for the header). */
/* Contigs of a tabix-indexed fragment file according to its .tbi (those holding rows, newline-separated,
 * NUL-terminated; *needed_out = bytes required) and the layout of its first data row -- enough to open a
 * file lazily and decode single contigs on demand (a single-contig ftk_fragstream_open seeks through the
 * index).  FTK_ERR_FORMAT when there is no usable index (callers then decode the file in one pass). */
int ftk_fragfile_index_contigs(const char* path, char* names_out, int64_t cap, int64_t* needed_out, int* is_bed6_out);
typedef struct ftk_fragstream ftk_fragstream;
/* ftk_fragstream_open_device: the same stream with the ROW PARSER ON THE GPU for text files (BAM records stay
 * on the host): the host threads only inflate, each piece of text goes to device `device_id` in one DMA, four
 * small kernels find the lines and parse the plain rows, and the tables handed out hold DEVICE columns
 * (ftk_fragtable_is_device; ftk_fragtable_columns then returns device pointers, ftk_fragtable_columns_to_host
 * copies them out).  A piece that holds anything but plain rows (comments, signs, blanks, short lines ...) is
 * parsed by the host's field-rule parser, so the rows are the same as ftk_fragstream_open's.  ftk_frags_from_table
 * takes both kinds.  FTK_DEVICE_PARSE=0 makes it behave like ftk_fragstream_open. */
int ftk_fragstream_open_device(int device_id, const char* path, const char* contig /* NULL = all */, int is_bam,
                               int n_threads, int max_queued, ftk_fragstream** out);

/* A stream over the rows of ONE REGION of a contig (the reference's per-window `fetch(contig, start, stop)`,
 * io/alignment.py:205-268, at the granularity a rank of a multi-GPU run needs: frag/_delfi.py's ranks each take a
 * window-aligned share of the genome).  The single table it hands out holds EVERY row of `contig` that overlaps
 * [start, stop) - for a BAM: every fragment whose read1 record overlaps it - and may hold more (rows before the region
 * from the first block read; the whole contig for files whose index has no linear index, streams without a device or
 * with FTK_DEVICE_INFLATE=0 / FTK_DEVICE_BAM_PARSE=0).  With a tabix index (a BAI for BAM input) the
 * read starts at the linear index's offset for `start` and ends where the parsed rows say the region is complete: a
 * row that starts at or behind `stop`, or another contig's rows, were seen (the index's 16 kb windows give a first
 * guess; a row longer than a window makes the read go on in 8 MB steps).  Arguments otherwise as
 * ftk_fragstream_open_device. */
int ftk_fragstream_open_region(int device_id, const char* path, const char* contig, int64_t start, int64_t stop, int is_bam,
                               int n_threads, int max_queued, ftk_fragstream** out);
int ftk_fragtable_is_device(const ftk_fragtable* t, int i);
/* hipEvent_t recorded behind the last write to a device table's columns (NULL for host tables) */
void* ftk_fragtable_ready_event(const ftk_fragtable* t, int i);
int ftk_fragtable_columns_to_host(const ftk_fragtable* t, int i, int32_t* start, int32_t* end, uint8_t* mapq,
                                  uint8_t* strand);
/* The read1 span and the file-order rank of a BAM table's rows into host arrays (any may be NULL), wherever the
 * table's columns live (BAM records parsed on the device hand out device columns: ftk_fragtable_is_device). */
int ftk_fragtable_read1_to_host(const ftk_fragtable* t, int i, int32_t* r1_start, int32_t* r1_end, int32_t* order);
int ftk_fragstream_open(const char* path, const char* contig /* NULL = all */, int is_bam, int n_threads,
                        int max_queued, ftk_fragstream** out);
int ftk_fragstream_next(ftk_fragstream* s, ftk_fragtable** out);
int ftk_fragstream_n_refs(ftk_fragstream* s);
const char* ftk_fragstream_ref_name(ftk_fragstream* s, int i);
int64_t ftk_fragstream_ref_length(ftk_fragstream* s, int i);
/* Wall time (ms) the stream's producer thread spent per stage, complete once ftk_fragstream_next has returned the
 * end of the file: out[0..5] = file read, BGZF inflate, row / record parse (device parser: launch), run merge /
 * collect, hand-over (pack + waiting for queue space), everything else.  The stages of one piece run one after
 * the other on the producer; their work is spread over n_threads (and the GPU for the device row parser). */
int ftk_fragstream_stage_ms(ftk_fragstream* s, double out[6]);
/* BAM records met so far (cumulative; read it after every ftk_fragstream_next, the final one included) that the
 * reference does NOT simply skip and this library cannot turn into a row (csrc/ftk_bamrule.h):
 *   out[0]  read1 records whose fragment the int32 columns cannot hold - a NEGATIVE start (reference_end + TLEN < 0,
 *           io/alignment.py:257) or a coordinate beyond 2^31 - 1.  The reference yields such a fragment; here it is
 *           dropped and counted (the Python surface issues a UserWarning).
 *   out[1]  read1 records WITHOUT a CIGAR and TLEN < 0.  pysam's reference_end is None for them and the reference
 *           raises TypeError at io/alignment.py:257; here they are dropped and counted (the Python surface raises
 *           TypeError).  A CIGAR-less read1 with TLEN > 0 is a fragment, as in the reference (:253-255).
 * ftk_fragtable_skipped: the same two numbers for a table of the whole-file decoder ftk_bam_decode. */
int ftk_fragstream_skipped(ftk_fragstream* s, int64_t out[2]);
int ftk_fragtable_skipped(const ftk_fragtable* t, int64_t out[2]);
void ftk_fragstream_close(ftk_fragstream* s);
const char* ftk_fragtable_error(void); /* message for a failed decode call (thread-local) */
int ftk_fragtable_is_bed6(const ftk_fragtable* t);
int ftk_fragtable_n_contigs(const ftk_fragtable* t);
const char* ftk_fragtable_contig_name(const ftk_fragtable* t, int i);
int64_t ftk_fragtable_contig_length(const ftk_fragtable* t, int i); /* BAM @SQ LN, -1 for text */
int64_t ftk_fragtable_contig_rows(const ftk_fragtable* t, int i);
int ftk_fragtable_columns(const ftk_fragtable* t, int i, const int32_t** start, const int32_t** end,
                          const uint8_t** mapq, const uint8_t** strand,
                          const int32_t** r1_start /* NULL for text */, const int32_t** r1_end);
/* 1 when contig i's columns sit in page-locked host memory (a HIP device was present at decode time). */
/* BAM tables: rank of each fragment's read1 record in the file (pysam iterates in that order, the
 * columns are sorted by fragment start); NULL for text tables, whose rows already are in file order. */
int ftk_fragtable_order(const ftk_fragtable* t, int i, const int32_t** order);
int ftk_fragtable_is_pinned(const ftk_fragtable* t, int i);
void ftk_fragtable_free(ftk_fragtable* t);

/* Page-locked host memory for large result arrays (the per-base scores of frag/_wps.py:181-188 are
 * 8 bytes a base): a device -> host copy into it is one DMA at PCIe speed, into pageable memory a staged
 * copy at a third to half of that.  Blocks are recycled by ftk_host_free (a fresh one is a 2 MB-page mapping touched by
 * a few threads and registered with the driver: ~0.01 ms per MB; hipHostMalloc, the fall-back, ~0.2 ms per MB).
 * FTK_ERR_NO_DEVICE without a HIP device; FTK_ERR_OOM when the driver refuses or more than 8 GB
 * (FTK_PINNED_RESULT_LIMIT_MB) would be outstanding - the caller then uses ordinary memory. */
int ftk_host_alloc(int64_t bytes, void** out);
/* The same recycling for a result the DEVICE never writes: ordinary (pageable) memory, 2 MB-aligned.  ftk_wps with a
 * host output of 4 M positions or more sends the scores across the link as int16 and the host threads widen them into
 * the output (see ftk_wps) - page-locking such an output buys nothing and took 0.2 ms per MB when this entry point was added (0.4 s for a chr1 of scores, paid
 * by the first call of a process = by every command-line call); the widening threads fault this one in in parallel.
 * Works without a device.  Freed with ftk_host_free like the page-locked blocks. */
int ftk_host_alloc_pageable(int64_t bytes, void** out);
void ftk_host_free(void* p);
/* Give back everything the library keeps for reuse between calls - idle page-locked blocks (decoded tables, result
 * arrays from ftk_host_alloc), idle device blocks of contigs parsed on the GPU, the streaming decoders' idle buffer
 * sets (up to four per process, ~1 GB of page-locked and ~2 GB of device memory after a large text stream).  Blocks
 * in use are not touched.  Returns the bytes released (a lower bound).  For long-lived processes between jobs; the
 * next call that needs a block allocates it again (~0.01 ms per MB for the page-locked ones). */
int64_t ftk_cache_trim(void);
/* Upload contig i of a decoded table (including the BAM read1 columns) as contig_id: the
 * decode -> pinned SoA -> hipMemcpyAsync leg of the pipeline, without a detour through the caller. */
int ftk_frags_from_table(ftk_ctx* ctx, int contig_id, const ftk_fragtable* t, int i);

/* File -> HBM in one call, for hosts that do not want to handle tables: the streaming decoder feeds
 * every contig of the file (or only `contig`) to ftk_frags_from_table as it is decoded; contig ids are
 * first_contig_id, first_contig_id + 1, ... in file order (BAM: contigs that have usable reads, in
 * header order), *n_loaded_out of them; ftk_frags_name gives the contig name behind an id loaded this
 * way.  (io/alignment.py:160-203 opens the same two kinds of input.) */
int ftk_frags_load_fraggz(ftk_ctx* ctx, const char* path, const char* contig /* NULL = all */, int n_threads,
                          int first_contig_id, int* n_loaded_out);
int ftk_frags_load_bam(ftk_ctx* ctx, const char* path, const char* contig /* NULL = all */, int n_threads,
                       int first_contig_id, int* n_loaded_out);
const char* ftk_frags_name(ftk_ctx* ctx, int contig_id);

/* ---- a5: coverage --------------------------------------------------------
 * frag/_coverage.py:117-130 (`for _ in frags: coverage += 1`) over
 * utils/_frag_generator.py:117-130, for n_win windows of one contig at once.
 * Windows may overlap and come in any order.  count_out[i] = number of
 * fragments passing `f` for window i. */
int ftk_window_counts(ftk_ctx* ctx, int contig_id, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                      const ftk_filter* f, int64_t* count_out);

/* ---- a10/a11: DELFI short/long counts ------------------------------------
 * frag/_delfi.py:443-472 per window: mapq >= mapq_min, 100 <= len <= 220,
 * midpoint in [w_start, w_end), not inside a blacklist region that lies fully
 * inside the window (frag/_delfi.py:110-126,455-462), not
 * ContigGaps.in_tcmere (genome/gaps.py:217-237; note the all() over
 * telomeres); len >= 151 -> long, else short; nfrag = short + long.
 * bl_start/bl_end: the contig's blacklist, sorted by (start, stop)
 * (frag/_delfi.py:85-107); may be NULL when n_bl == 0.  The window-level
 * NOARM gate (frag/_delfi.py:423-428) is the caller's.  Windows and blacklist
 * are HOST arrays; their device form (windows + per-window blacklist CSR) is
 * cached in the ctx by content, so repeated calls with the same bins upload
 * nothing.  Outputs may be host or device pointers; nfrag_out may be NULL. */
int ftk_delfi_counts(ftk_ctx* ctx, int contig_id, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                     int32_t mapq_min, const int32_t* bl_start, const int32_t* bl_end, int64_t n_bl,
                     const ftk_gaps* gaps, int64_t* short_out, int64_t* long_out, int64_t* nfrag_out);

/* ---- a9: fragment-length histogram per window -----------------------------
 * frag/_frag_length.py:147-153 (_distribution_from_gen) for n_win windows:
 * hist_out[i * n_bins + (len - len_lo)] counts fragments of window i passing
 * `f`; lengths outside [len_lo, len_lo + n_bins) are tallied in
 * overflow_out[i].  The statistics of frag/_frag_length.py:175-238 are host
 * arithmetic on these histograms. */
int ftk_fraglen_hist(ftk_ctx* ctx, int contig_id, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                     const ftk_filter* f, int32_t len_lo, int32_t n_bins, uint32_t* hist_out, int64_t* overflow_out);
/* a9, the statistics themselves (frag/_frag_length.py:156-172 `_find_median`, :202-224 `_frag_length_stats`): the same
 * pass as ftk_fraglen_hist over lengths [len_lo, len_lo + n_bins), then per window - on the device, from the histogram
 * rows, which never leave it - stats_out[w][7] = mean, median, stdev (population), min, max, count, #(len <= short_cut)
 * as float64 (the integers are exact).  count 0: the window holds no passing fragment (the other six are 0; the
 * reference reports -1 for all).  The length range must hold every length the filter passes (the caller takes
 * [max(min_len, 0), min(max_len, longest fragment of the contig)]); n_bins as in ftk_fraglen_hist.  stats_out may be
 * host or device memory. */
int ftk_fraglen_stats(ftk_ctx* ctx, int contig_id, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                      const ftk_filter* f, int32_t len_lo, int32_t n_bins, int32_t short_cut, double* stats_out);

/* ---- fused pass: any combination of the three window features in ONE sweep over
 * the contig's fragments (one plan, one launch pair instead of three): coverage
 * (count_out) and length histogram (hist_out/overflow_out) under `f`, DELFI
 * short/long under the rules of ftk_delfi_counts.  A NULL output switches its
 * feature off; windows and blacklist must be host arrays when DELFI is on. */
int ftk_window_features(ftk_ctx* ctx, int contig_id, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                        const ftk_filter* f, int64_t* count_out, int32_t len_lo, int32_t n_bins, uint32_t* hist_out,
                        int64_t* overflow_out, int32_t delfi_mapq_min, const int32_t* bl_start, const int32_t* bl_end,
                        int64_t n_bl, const ftk_gaps* gaps, int64_t* short_out, int64_t* long_out);

/* The same pass for the windows of SEVERAL contigs in ONE launch (a genome-wide bin set: the
 * reference's delfi() / coverage() loop over contigs, frag/_delfi.py:289-300, frag/_coverage.py:244-248).
 * Item i holds contig items[i].contig_id's windows (host arrays; blacklist / gaps only matter for the
 * DELFI outputs).  Outputs are concatenated in item order: row = sum of n_win of the items before + w.
 * One block per window: meant for bin tilings (many windows of similar length). */
typedef struct ftk_feature_item {
    int32_t contig_id;
    int64_t n_win;
    const int32_t* w_start;
    const int32_t* w_end;
    const int32_t* bl_start; /* sorted by start; NULL = no blacklist */
    const int32_t* bl_end;
    int64_t n_bl;
    const ftk_gaps* gaps;    /* NULL = no gap annotation */
} ftk_feature_item;
int ftk_window_features_batch(ftk_ctx* ctx, const ftk_feature_item* items, int32_t n_items, const ftk_filter* f,
                              int64_t* count_out, int32_t len_lo, int32_t n_bins, uint32_t* hist_out,
                              int64_t* overflow_out, int32_t delfi_mapq_min, int64_t* short_out, int64_t* long_out);

/* frag/_frag_length.py:290-305 (frag_length): lengths of the fragments of ONE
 * window passing `f`, in file order.  Writes at most cap values to len_out
 * and always the true count to n_out. */
int ftk_frag_lengths(ftk_ctx* ctx, int contig_id, int32_t w_start, int32_t w_end, const ftk_filter* f,
                     int32_t* len_out, int64_t cap, int64_t* n_out);

/* utils/utils.py:186-255 (frag_array) / utils/_frag_generator.py:58-141: the
 * passing fragments of ONE window in file order (any column pointer may be
 * NULL). */
int ftk_frag_select(ftk_ctx* ctx, int contig_id, int32_t w_start, int32_t w_end, const ftk_filter* f,
                    int32_t* start_out, int32_t* end_out, uint8_t* mapq_out, uint8_t* strand_out, int64_t cap,
                    int64_t* n_out);

/* ---- a7/a8: Windowed Protection Score --------------------------------------
 * frag/_wps.py:25-53,156-188 for every base c in [start, stop):
 *   ws = rint(c - W/2), we = rint(c + W/2 - 1)   (numpy rint: half to even)
 *   WPS(c) = #{fs < ws and fe > we} - #{ws <= fs <= we or ws <= fe <= we}
 * over the fragments with mapq >= mapq_min, min_len <= len <= max_len whose
 * midpoint lies in [max(start - max_len, 0), min(stop + max_len, chrom_size))
 * (frag/_wps.py:156-169).  wps_out has stop - start entries. */
int ftk_wps(ftk_ctx* ctx, int contig_id, int64_t start, int64_t stop, int64_t chrom_size, int32_t window_size,
            int32_t min_len, int32_t max_len, int32_t mapq_min, int64_t* wps_out);
/* frag/_multi_wps.py:196-198: the same for n_iv intervals of one contig in
 * one launch; interval i writes iv_stop[i] - iv_start[i] values at
 * wps_out + out_offset[i].  iv_* / out_offset are host arrays. */
int ftk_wps_intervals(ftk_ctx* ctx, int contig_id, const int64_t* iv_start, const int64_t* iv_stop, int64_t n_iv,
                      const int64_t* out_offset, int64_t chrom_size, int32_t window_size, int32_t min_len,
                      int32_t max_len, int32_t mapq_min, int64_t* wps_out);

/* ftk_wps with the copy-back taken off the caller's path: returns once the kernel (ctx stream) and the copy of
 * the scores into wps_out_host (the ctx's copy stream, behind the kernel) are enqueued; the next calls on the ctx
 * - loading and scoring the next contig - overlap the copy.  *token_out identifies the result: the array is
 * valid after ftk_result_wait(ctx, token) (or ftk_ctx_sync).  Two results can be in flight; a third call
 * waits for the older one's copy.  Tokens count up and are never reused: waiting for an old token whose buffer has
 * been taken over by a later call returns at once (that call waited for the old copy before it reused the buffer); a
 * token that was never handed out is FTK_ERR_INVALID.  wps_out_host should come from ftk_host_alloc (a pageable array
 * is copied through a staging buffer and gains nothing).  A degenerate interval gives token -1 (nothing to wait for). */
int ftk_wps_async(ftk_ctx* ctx, int contig_id, int64_t start, int64_t stop, int64_t chrom_size, int32_t window_size,
                  int32_t min_len, int32_t max_len, int32_t mapq_min, int64_t* wps_out_host, int* token_out);
int ftk_result_wait(ftk_ctx* ctx, int token);

/* ftk_window_features (all arguments as there) FOLLOWED BY ftk_wps (all arguments as there) on the same contig, as
 * ONE launch when the feature request takes the FAST block path (midpoint policy, no length bounds on the coverage
 * filter, one mapq cut, a bin tiling; tabix fetch or - on a contig with read1 columns - the BAM read1 fetch rule):
 * the grid holds the feature blocks first and the WPS tiles behind them, so the feature pass's tail and the WPS ramp
 * overlap and WPS finds the contig's columns in the Infinity Cache.  wps_out may be device memory or a host array
 * (the scores then cross the link like ftk_wps': 16 bits each when they fit).  Any other request runs as the two
 * launches.  Results are those of the two calls (reference: frag/_coverage.py:117-130, _frag_length.py:147-153,
 * _delfi.py:443-472, _wps.py:156-188; BAM fetch rule: io/alignment.py:242-268). */
int ftk_window_features_wps(ftk_ctx* ctx, int contig_id, const int32_t* w_start, const int32_t* w_end, int64_t n_win,
                            const ftk_filter* f, int64_t* count_out, int32_t len_lo, int32_t n_bins, uint32_t* hist_out,
                            int64_t* overflow_out, int32_t delfi_mapq_min, const int32_t* bl_start, const int32_t* bl_end,
                            int64_t n_bl, const ftk_gaps* gaps, int64_t* short_out, int64_t* long_out, int64_t start,
                            int64_t stop, int64_t chrom_size, int32_t window_size, int32_t min_len, int32_t max_len,
                            int32_t mapq_min, int64_t* wps_out);

/* WPS of a whole contig AND the window features of a regular bin tiling in ONE pass over the fragments
 * (BASELINE config 5: coverage + WPS + length histogram + DELFI fused): bin k = [win_start + k * win_len,
 * win_start + (k + 1) * win_len), k < n_win, midpoint policy, tabix fetch semantics.  The block that scores
 * a 4096-base tile also classifies the fragments starting in it, so [start, stop) must cover every
 * fragment start of the contig (start <= 0, stop > last start) and win_len must be at least 4096 + the
 * longest fragment.  Outputs as in ftk_wps / ftk_window_features (NULL switches a feature off; results are
 * identical to the separate calls). */
int ftk_wps_window_features(ftk_ctx* ctx, int contig_id, int64_t start, int64_t stop, int64_t chrom_size,
                            int32_t window_size, int32_t min_len, int32_t max_len, int32_t mapq_min, int64_t* wps_out,
                            int32_t win_start, int32_t win_len, int32_t n_win, const ftk_filter* f, int64_t* count_out,
                            int32_t len_lo, int32_t n_bins, uint32_t* hist_out, int64_t* overflow_out,
                            int32_t delfi_mapq_min, const int32_t* bl_start, const int32_t* bl_end, int64_t n_bl,
                            const ftk_gaps* gaps, int64_t* short_out, int64_t* long_out);

/* Intervals of SEVERAL contigs in one launch: interval i is [iv_start[i], iv_stop[i]) of contig
 * contig_ids[i] (length chrom_size[i]) and writes its scores at wps_out + out_offset[i].  All arrays
 * except wps_out (host or device) are host arrays of n_iv entries (n_iv <= 64: whole contigs or large ranges). */
int ftk_wps_batch(ftk_ctx* ctx, const int32_t* contig_ids, const int64_t* iv_start, const int64_t* iv_stop,
                  const int64_t* chrom_size, const int64_t* out_offset, int64_t n_iv, int32_t window_size,
                  int32_t min_len, int32_t max_len, int32_t mapq_min, int64_t* wps_out);

/* ---- next row: cleavage profile ----------------------------------
 * frag/_cleavage_profile.py:33-90,204-216 for every base of [start, stop) (already
 * expanded by left/right and clipped by the caller, :201-202): fragments selected
 * like frag_array(start, stop, intersect_policy="any") with mapq >= mapq_min and
 * min_len <= len <= max_len (FTK_LEN_OPEN = no bound); depth = fragments covering the
 * base, ends = + fragments starting there plus - fragments whose stop is there;
 * prop_out = ends / depth * 100 (float64, 0 where depth == 0).
 * ftk_cleavage_intervals: n_iv intervals of one contig in one launch (host interval
 * arrays), interval i written at prop_out + out_offset[i]. */
int ftk_cleavage(ftk_ctx* ctx, int contig_id, int64_t start, int64_t stop, int32_t min_len, int32_t max_len,
                 int32_t mapq_min, double* prop_out);
int ftk_cleavage_intervals(ftk_ctx* ctx, int contig_id, const int64_t* iv_start, const int64_t* iv_stop, int64_t n_iv,
                           const int64_t* out_offset, int32_t min_len, int32_t max_len, int32_t mapq_min,
                           double* prop_out);

/* ---- next row: WPS post-processing (adjust_wps) ---------------------
 * frag/_adjust_wps.py:25-50,119-140 for n_iv score runs laid end to end in `scores`
 * (run i = scores[offsets[i] .. offsets[i+1]), offsets is a host array of n_iv + 1):
 *   x      = scores - edge_sub[i]                      (edge_sub NULL = no subtraction; host array)
 *   adj[o] = x[o + W/2] - stat(x[o .. o+W))            o in [0, len_i - W), stat = median, or mean
 *                                                      when use_mean; W = median_window, even, 2..2048
 *   out    = adj, or savgol_filter(adj, savgol_window, mode="interp") when savgol_window > 0:
 *            interior o: sum_j coef[j] * adj[o - h + j]            (h = savgol_window / 2, window odd)
 *            first h   : sum_j edge[o][j]     * adj[j]             (polynomial edge fit as a matrix)
 *            last h    : sum_j edge[h + k][j] * adj[m - window + j], o = m - h + k
 *            coef [savgol_window] and edge [2*h][savgol_window] are host arrays prepared by the caller.
 * Run i writes len_i - W values at out + offsets[i] - i*W.  Every run needs len_i >= W and, with the
 * Savitzky-Golay pass, len_i - W >= savgol_window (the reference raises ValueError in both cases).
 * The median is exact (selection, no arithmetic besides the final (a+b)/2). */
int ftk_wps_adjust(ftk_ctx* ctx, const double* scores, const int64_t* offsets, int64_t n_iv, int32_t median_window,
                   int use_mean, const double* edge_sub, int32_t savgol_window, const double* savgol_coef,
                   const double* savgol_edge, double* out);

/* ---- next row: DELFI per-bin GC count on the device ------------------
 * frag/_delfi.py:476-490 counts G + C of the upper-cased window sequence
 * (io/reference.py:120-189) once per bin on the host.  Here a contig's reference image
 * is uploaded once and every bin is counted in one launch:
 *   FTK_REF_FASTA_TEXT  the contig's raw FASTA text (bases and line breaks, no header line);
 *                       ranges are BYTE offsets into the image (faidx arithmetic is the caller's);
 *   FTK_REF_2BIT        the packed DNA of a .2bit record (T=0 C=1 A=2 G=3, first base in the high
 *                       bits); ranges are BASE positions; N / mask blocks are the caller's to subtract.
 * gc_out[i] = number of G/C (either case) in [range_lo[i], range_hi[i]). */
#define FTK_REF_FASTA_TEXT 0
#define FTK_REF_2BIT 1
int ftk_ref_upload(ftk_ctx* ctx, int ref_id, const uint8_t* image, int64_t n_bytes, int kind);
/* The same image read straight from a file: n_bytes at file_offset of `path` (the packed DNA of a 2bit record, the
 * text of a FASTA record), through page-locked chunks on several read threads, copies asynchronous on the ctx
 * stream.  Device blocks of released images are recycled (no device-wide synchronisation per contig). */
int ftk_ref_upload_file(ftk_ctx* ctx, int ref_id, const char* path, int64_t file_offset, int64_t n_bytes, int kind);
int ftk_ref_release(ftk_ctx* ctx, int ref_id);
int ftk_ref_gc_counts(ftk_ctx* ctx, int ref_id, const int64_t* range_lo, const int64_t* range_hi, int64_t n,
                      int64_t* gc_out);

/* ---- next row: end-motif / breakpoint-motif k-mer histograms ----------
 * frag/_end_motifs.py:51-187 (region_end_motifs) and frag/_breakpoint_motifs.py:53-196
 * (region_breakpoint_motifs) per window: every fragment the index query returns for the window
 * (overlap, or read1 overlap for BAM, and mapq >= mapq_min -- the reference applies NO length
 * filter here) adds
 *    forward: the k-mer ref[start + fwd_offset, +k)
 *    reverse: the reverse complement of ref[stop + rev_offset, +k)
 * to the window's histogram of 4^k bins in ACGT order (utils/utils.py:388-410, gen_kmers).
 *    both_strands            both k-mers of every fragment
 *    else negative_strand    only the reverse k-mer, of every fragment
 *    else                    only the forward k-mer, of forward-strand fragments
 * A k-mer holding anything but A/C/G/T (either case) is not counted.  A forward k-mer that falls
 * off the contig drops the fragment; a reverse one that does is counted in err_out when
 * rev_oob_is_error (the reference raises RuntimeError there, _end_motifs.py:137-145) and skipped
 * otherwise.  guard > 0 drops fragments with start - guard < 0 or start + guard >= chrom_len
 * (_breakpoint_motifs.py:127-135).
 *   end motifs:        k,  fwd_offset 0,     rev_offset -k,    guard 0,   rev_oob_is_error = both_strands
 *   breakpoint motifs: k (even), fwd_offset -k/2, rev_offset -k/2, guard k/2, rev_oob_is_error 0
 * The reference image needs its geometry first (ftk_ref_set_layout): contig length, and the FASTA
 * line layout (bases per line, bytes per line) or the 2bit record's N blocks (host arrays). */
typedef struct ftk_motif {
    int32_t k; /* 1..7 */
    int32_t fwd_offset, rev_offset;
    int32_t both_strands, negative_strand;
    int32_t guard;
    int32_t rev_oob_is_error;
} ftk_motif;
int ftk_ref_set_layout(ftk_ctx* ctx, int ref_id, int64_t chrom_len, int32_t line_bases, int32_t line_width,
                       const int32_t* nblock_start, const int32_t* nblock_end, int64_t n_nblocks);
int ftk_motif_counts(ftk_ctx* ctx, int contig_id, int ref_id, const int32_t* w_start, const int32_t* w_end,
                     int64_t n_win, const ftk_motif* motif, int32_t mapq_min, int32_t fetch_mode,
                     uint32_t* counts_out /* [n_win][4^k] */, int64_t* nfrag_out /* [n_win] or NULL */,
                     int64_t* err_out /* [n_win] */);

/* ---- BGZF inflate on the device -------------------------------------------------------------------
 * The streaming decoder's host threads spend most of a fragment file's decode in DEFLATE; BGZF blocks are
 * independent streams of at most 64 KB of data, decoded here one wavefront per block (csrc/ftk_inflate.hip).
 * This entry point inflates a whole BGZF image (every block a BGZF member, as bgzip / htslib write them) held
 * in host memory and returns the data in host memory: *n_out = sum of the blocks' ISIZE fields; FTK_ERR_INVALID
 * when cap is too small (with *n_out set), FTK_ERR_FORMAT for anything that is not BGZF or does not decode to
 * its ISIZE.  An image of up to 4 GB of data (compressed and inflated) per call; compressed bytes beyond 2^28 go in
 * further launches of the kernel, which addresses its input by 32-bit bit positions.  (The fragment stream uses the
 * same kernel on device-resident pieces: FTK_DEVICE_INFLATE.) */
int ftk_bgzf_inflate_device(ftk_ctx* ctx, const uint8_t* file_bytes, int64_t n, uint8_t* out, int64_t cap, int64_t* n_out);

/* ---- output writers (host only; no ctx / GPU needed) ---------------------------------------------
 * The reference prints per-base results one Python f-string at a time (frag/_wps.py:208-229 WIG,
 * frag/_multi_wps.py:328-341 bedGraph) and hands bigWig entries to pyBigWig (:300-325).  These format the
 * same bytes on all host threads.  Buffers and arrays returned through `out` pointers are owned by the
 * library until ftk_buffer_free.  n_threads <= 0: all usable cores.  Errors: ftk_fragtable_error().
 * Several results are laid end to end as RUNS: run k = values[offsets[k] .. offsets[k+1]) starts at base
 * iv_start[k] (what ftk_wps_intervals / ftk_cleavage_intervals produce; one run = one interval).
 *   ftk_format_wig_i64       "<v>\n" per value (the body of a fixedStep WIG)
 *   ftk_format_bedgraph_i64  "<contig>\t<pos>\t<pos+1>\t<v>\n" per value of every run
 *   ftk_format_bedgraph_f64  the same for float64 values printed as Python's repr(float) prints them
 *                            (shortest round-trip digits; exponent form outside 1e-4 <= |v| < 1e16)
 *   ftk_file_write           data -> path (truncate or append); gzip_level > 0 writes gzip members of 1 MB
 *                            of text each, compressed in parallel (a valid multi-member .gz: gzip.open,
 *                            zcat, bgzip -d read it as one stream)
 *   ftk_gzip_members         the same gzip members into memory (*out, ftk_buffer_free): what a rank that does
 *                            not own the output file hands to the rank that writes it (multi-GPU runs of
 *                            multi_wps / multi_cleavage_profile: every rank formats and compresses the rows of
 *                            its own contigs, rank 0 concatenates the members in order)
 *   ftk_bigwig_fixedstep_sections  the data sections of one `addEntries(chrom, start, values=..., span=1,
 *                            step=1)` call PER RUN: values (value_kind 0 = int64, 1 = float64) cut into
 *                            sections of items_per_section float32 items (never across runs), each with
 *                            its 24-byte section header, zlib-compressed; *out holds the sections back to
 *                            back; section_table_out[3s..3s+2] = chromStart, chromEnd, compressed bytes of
 *                            section s; section_stats_out[4s..4s+3] = min, max, sum, sum of squares of its
 *                            float32 values (inputs of the zoom level and the total summary). */
int ftk_format_wig_i64(const int64_t* values, int64_t n, int n_threads, char** out, int64_t* out_len);
int ftk_format_bedgraph_i64(const char* contig, const int64_t* iv_start, const int64_t* offsets, int64_t n_iv,
                            const int64_t* values, int n_threads, char** out, int64_t* out_len);
int ftk_format_bedgraph_f64(const char* contig, const int64_t* iv_start, const int64_t* offsets, int64_t n_iv,
                            const double* values, int n_threads, char** out, int64_t* out_len);
void ftk_buffer_free(void* p);
int ftk_file_write(const char* path, const char* data, int64_t n, int gzip_level, int n_threads, int append);
int ftk_gzip_members(const char* data, int64_t n, int gzip_level, int n_threads, char** out, int64_t* out_len);
/* The array frag/_wps.py:181-188 returns -- numpy records ('contig', 'U16'), ('start', 'i8'), ('wps', 'i8'), 80 bytes
 * each: contig name as 16 UCS-4 code points (zero padded), start + i, values[i] -- filled by the host threads (for a
 * chromosome that array is gigabytes; one numpy thread takes longer over it than the GPU over the scores). */
int ftk_fill_wps_records(void* dst, int64_t n, const uint32_t contig_ucs4[16], int64_t start, const int64_t* values,
                         int n_threads);
/* Fragment-file output (what the decoders read, io/alignment.py:270-302): rows "<contig>\t<start>\t<end>\t<mapq>\t<+|->\n"
 * (bed6: a "." name column before mapq), and a BGZF container writer (blocks of 0xFF00 bytes of data compressed in
 * parallel, the standard EOF block when write_eof): block_offsets[k] = file offset of the block holding data bytes
 * [k * 0xFF00, ...), block_offsets[n_blocks] = offset behind the last data block -- what a tabix / BAI index needs. */
int ftk_format_frag_rows(const char* contig, const int32_t* start, const int32_t* end, const uint8_t* mapq,
                         const uint8_t* strand, int64_t n, int bed6, int n_threads, char** out, int64_t* out_len);
int ftk_bgzf_write(const char* path, const char* data, int64_t n, int level, int n_threads, int append, int write_eof,
                   int64_t* block_offsets);
/* Test / bench tooling (BASELINE config 5 reads a whole-genome 60x BAM: ~1.2 G records): appends to `path` -- which
 * already holds the BAM header's BGZF blocks -- the coordinate-sorted paired-end records of ONE contig, two per
 * fragment (read1: flag 99 / 83, at the fragment's start when strand != 0, else at its end; the mate mirrored;
 * TLEN = +-(end - start); one `read_len`M CIGAR; name = the fragment's index in `name_len - 1` decimal digits), built,
 * sorted (position, read1 before read2, fragment index) and deflated by the host threads in windows of 512 kb, so memory
 * stays bounded and the rate is libdeflate's.  Fragments must be sorted by start with end >= start + read_len.
 * linear (may be NULL): the contig's 16 kb BAI linear index, n_linear >= ((contig_len + read_len) >> 14) + 1 entries
 * preset to all-ones; entry w receives the virtual offset of the first record overlapping window w.  first_off / end_off:
 * file offsets of the contig's first block and behind its last one (-1 / unchanged file end when it has no records).
 * What the reference reads these files with: pysam.AlignmentFile.fetch, io/alignment.py:242-268. */
int ftk_synth_bam_contig(const char* path, int32_t ref_id, int64_t contig_len, const int32_t* start, const int32_t* end,
                         const uint8_t* mapq, const uint8_t* strand, int64_t n, int32_t read_len, int32_t name_len,
                         uint64_t seed, int level, int n_threads, uint64_t* linear, int64_t n_linear,
                         int64_t* first_off, int64_t* end_off, int64_t* n_records_out);
int ftk_bigwig_fixedstep_sections(uint32_t chrom_id, const int64_t* iv_start, const int64_t* offsets, int64_t n_iv,
                                  const void* values, int value_kind, int32_t items_per_section, int level,
                                  int n_threads, char** out, int64_t* out_len, int64_t* n_sections_out,
                                  int64_t** section_table_out, double** section_stats_out);

/* ---- multi-GPU (SURVEY 8-e, 8-b's export list) -------------------------------------------------
 * One process and one ftk_ctx per GPU.  Contigs / genome runs are dealt to the ranks by the host and every window,
 * bin and base depends on its own contig's fragments only, so the data path has no collective.  What remains are the
 * two exchanges that stand where the reference collects its Pool's results: the all-gather of the per-bin DELFI /
 * per-interval rows (frag/_delfi.py:289-300: pool.starmap's result list; frag/_coverage.py:212-248) and the all-reduce
 * of the genome-wide total of coverage(normalize=True) (frag/_coverage.py:215-227) -- RCCL over xGMI, here behind the
 * C ABI so that a host in any language can shard (finaletoolkit_amd/comm.py is the ctypes host).  librccl is resolved
 * when the first communicator is created (dlopen "librccl.so.1": a copy already in the process, e.g. torch's, is
 * re-used); one-GPU hosts never load it.
 *
 * ftk_comm_create: rank `rank` of `world` joins the job's communicator on ctx's device.  `id_hex_or_path` is what the
 * ranks have in common: the 256 hex digits of an RCCL unique id (ftk_comm_unique_id on one rank, handed to the others
 * by the host's own means), or a FILE PATH all ranks can reach -- rank 0 creates the id and writes it there (atomic
 * rename), the others wait for it (FTK_COMM_TIMEOUT_S, default 600), rank 0 removes it as soon as the communicator is
 * up (every rank has read it by then).  RCCL's start-up banner is kept off stdout (FTK_COMM_BANNER=1 lets it through).
 * NULL is accepted for world == 1.  A collective runs on the communicator's own HIP stream behind the work the ctx stream
 * holds at the call, so later launches on the ctx stream overlap it.  Buffers may be host or device memory: with a
 * host buffer the call returns when the result is there; with device buffers it returns at once and ftk_comm_join
 * makes the ctx stream wait for the result (no host wait).  Every rank must make the same calls in the same order. */
typedef struct ftk_comm ftk_comm;
int ftk_comm_unique_id(char* hex_out /* [257] */);
int ftk_comm_create(ftk_ctx* ctx, int rank, int world, const char* id_hex_or_path, ftk_comm** out);
int ftk_comm_size(const ftk_comm* comm, int* rank_out, int* world_out);
/* recv[r * n + i] = rank r's send[i]; n equal on all ranks */
int ftk_allgather_i64(ftk_comm* comm, const int64_t* send, int64_t n, int64_t* recv);
/* values[i] = sum over ranks, in place */
int ftk_allreduce_sum_i64(ftk_comm* comm, int64_t* values, int64_t n);
/* point to point (the compressed output sections a rank hands to the writing rank, frag/_multi_wps.py:300-341's
 * parent-side writer): matching send / recv pairs, any size */
int ftk_comm_send(ftk_comm* comm, int dst, const void* data, int64_t n_bytes);
int ftk_comm_recv(ftk_comm* comm, int src, void* data, int64_t n_bytes);
int ftk_comm_join(ftk_comm* comm);
void ftk_comm_destroy(ftk_comm* comm);

#ifdef __cplusplus
}
#endif
#endif /* FTK_H */
