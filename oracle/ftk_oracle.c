/*
 * ftk_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C restatement of FinaleToolkit's per-window hot path, used only as the
 * parity checker (tests/, __graft_entry__.smoke) and as the timed CPU baseline
 * (bench.py cpu_baseline, kind "port").  Nothing under finaletoolkit_amd/ may
 * call it.  Pinned against the reference itself: oracle/gen_golden.py imports
 * the reference (through oracle/refstub.py, build container only) and writes
 * tests/golden/, which tests/test_oracle_golden.py replays against this file.
 *
 * Every function follows the reference loop it cites (paths relative to the
 * reference checkout, src/finaletoolkit/...): one window at a time, "fetch"
 * the rows an index query would return, then the per-fragment Python
 * predicate.  Input is one contig's fragments in file order (start-sorted, as
 * a tabix-indexed file is).  The index query is emulated with a bisection on
 * the sorted starts plus an explicit overlap test, so no candidate is missed:
 * a row overlapping [ws, we) has start < we and start > ws - max_len.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define OPEN_LO INT32_MIN
#define OPEN_HI INT32_MAX

typedef struct {
    int32_t mapq_min;
    int32_t min_len; /* -1 = None */
    int32_t max_len; /* -1 = None */
    int32_t policy;  /* 0 midpoint, 1 any, 2 none: whatever the index query returned (io/alignment.py:216-240) */
    int32_t fetch_mode; /* 0 tabix rows, 1 BAM read1 alignments */
} orc_filter;

typedef struct {
    int32_t has_gaps;
    int32_t cen_start, cen_stop;
    int32_t n_telo;
    int32_t telo_start[8];
    int32_t telo_stop[8];
} orc_gaps;

typedef struct {
    const int32_t* start;
    const int32_t* end;
    const uint8_t* mapq;
    const uint8_t* strand;
    const int32_t* r1s; /* BAM: read1 alignment span, else NULL */
    const int32_t* r1e;
    int64_t n;
    int32_t max_len; /* bisection bound: how far behind its start a row's tested interval (the row; BAM: read1) ends */
    int32_t hi_slack; /* BAM: how far in FRONT of its fragment's start a read1 alignment begins (0 for tabix rows) */
} orc_frags;

/* Python's `//` by 2 (floor), also for the negative sums a BAM fragment with a negative start gives. */
static int64_t floor_half(int64_t v) { return (v - (v < 0 ? 1 : 0)) / 2; }

static int64_t lower_bound_i32(const int32_t* a, int64_t n, int64_t v) {
    int64_t lo = 0, hi = n;
    while (lo < hi) {
        int64_t m = (lo + hi) / 2;
        if ((int64_t)a[m] < v) lo = m + 1; else hi = m;
    }
    return lo;
}

void orc_frags_init(orc_frags* f, const int32_t* start, const int32_t* end, const uint8_t* mapq,
                    const uint8_t* strand, const int32_t* r1s, const int32_t* r1e, int64_t n) {
    f->start = start; f->end = end; f->mapq = mapq; f->strand = strand; f->r1s = r1s; f->r1e = r1e; f->n = n;
    int64_t m = 0, h = 0;
    for (int64_t i = 0; i < n; ++i) {
        int64_t len = (int64_t)end[i] - start[i];
        if (len > m) m = len;
        if (r1s) { /* read1 need not lie inside its fragment (TLEN is whatever the aligner wrote) */
            if ((int64_t)r1e[i] - start[i] > m) m = (int64_t)r1e[i] - start[i];
            if ((int64_t)start[i] - r1s[i] > h) h = (int64_t)start[i] - r1s[i];
        }
    }
    f->max_len = (int32_t)(m > INT32_MAX ? INT32_MAX : m);
    f->hi_slack = (int32_t)(h > INT32_MAX ? INT32_MAX : h);
}

/* Row range that can contain anything an index query for [ws, we) returns. */
static void fetch_range(const orc_frags* f, int32_t ws, int32_t we, int64_t* lo, int64_t* hi) {
    *lo = (ws == OPEN_LO) ? 0 : lower_bound_i32(f->start, f->n, (int64_t)ws - f->max_len);
    *hi = (we == OPEN_HI) ? f->n : lower_bound_i32(f->start, f->n, (int64_t)we + f->hi_slack);
    if (*hi < *lo) *hi = *lo;
}

/* Was row i returned by the reference's fetch(contig, ws, we)?
 * tabix (io/alignment.py:273-279): row.start < we and row.end > ws.
 * BAM   (io/alignment.py:245): read1 alignment overlaps the window. */
static int fetched(const orc_frags* f, int64_t i, int32_t ws, int32_t we, int fetch_mode) {
    if (fetch_mode == 1)
        return (we == OPEN_HI || f->r1s[i] < we) && (ws == OPEN_LO || f->r1e[i] > ws);
    return (we == OPEN_HI || f->start[i] < we) && (ws == OPEN_LO || f->end[i] > ws);
}

/* utils/_frag_generator.py:117-130 with io/alignment.py:291 (mapq) */
static int passes(const orc_frags* f, int64_t i, int32_t ws, int32_t we, const orc_filter* flt) {
    int32_t fs = f->start[i], fe = f->end[i];
    if ((int32_t)f->mapq[i] < flt->mapq_min) return 0;            /* io/alignment.py:291 */
    int32_t len = fe - fs;                                        /* io/alignment.py:51-54 */
    if (flt->min_len != -1 && !(len >= flt->min_len)) return 0;   /* _comparison.py:20-24 */
    if (flt->max_len != -1 && !(len <= flt->max_len)) return 0;   /* _comparison.py:13-17 */
    if (flt->policy == 0) {                                       /* _frag_generator.py:35-42 */
        int64_t mid = floor_half((int64_t)fs + (int64_t)fe);
        if (ws != OPEN_LO && !(mid >= ws)) return 0;
        if (we != OPEN_HI && !(mid < we)) return 0;
    } else if (flt->policy == 1) {                                /* _frag_generator.py:44-50 */
        if (ws != OPEN_LO && !(fe > ws)) return 0;
        if (we != OPEN_HI && !(fs < we)) return 0;
    }
    return 1;
}

/* frag/_coverage.py:117-130: `for _ in frags: coverage += 1`, per window */
void orc_window_counts(const orc_frags* f, const int32_t* ws, const int32_t* we, int64_t n_win,
                       const orc_filter* flt, int64_t* count_out) {
    for (int64_t w = 0; w < n_win; ++w) {
        int64_t lo, hi, cov = 0;
        if (we[w] < ws[w]) { count_out[w] = 0; continue; }
        fetch_range(f, ws[w], we[w], &lo, &hi);
        for (int64_t i = lo; i < hi; ++i)
            if (fetched(f, i, ws[w], we[w], flt->fetch_mode) && passes(f, i, ws[w], we[w], flt)) cov += 1;
        count_out[w] = cov;
    }
}

/* frag/_frag_length.py:147-153 (_distribution_from_gen): dict length -> count,
 * held here as a dense histogram over [len_lo, len_lo + n_bins). */
void orc_fraglen_hist(const orc_frags* f, const int32_t* ws, const int32_t* we, int64_t n_win,
                      const orc_filter* flt, int32_t len_lo, int32_t n_bins, uint32_t* hist_out,
                      int64_t* overflow_out) {
    memset(hist_out, 0, (size_t)n_win * (size_t)n_bins * sizeof(uint32_t));
    for (int64_t w = 0; w < n_win; ++w) {
        int64_t lo, hi;
        overflow_out[w] = 0;
        if (we[w] < ws[w]) continue;
        fetch_range(f, ws[w], we[w], &lo, &hi);
        for (int64_t i = lo; i < hi; ++i) {
            if (!(fetched(f, i, ws[w], we[w], flt->fetch_mode) && passes(f, i, ws[w], we[w], flt))) continue;
            int32_t b = (f->end[i] - f->start[i]) - len_lo;
            if (b >= 0 && b < n_bins) hist_out[(size_t)w * n_bins + b] += 1; else overflow_out[w] += 1;
        }
    }
}

/* utils/_frag_generator.py:117-130 stream for ONE window, in file order.
 * Returns the number of passing rows; writes at most cap of them. */
int64_t orc_frag_select(const orc_frags* f, int32_t ws, int32_t we, const orc_filter* flt, int32_t* start_out,
                        int32_t* end_out, uint8_t* mapq_out, uint8_t* strand_out, int64_t cap) {
    int64_t lo, hi, k = 0;
    if (we < ws) return 0;
    fetch_range(f, ws, we, &lo, &hi);
    for (int64_t i = lo; i < hi; ++i) {
        if (!(fetched(f, i, ws, we, flt->fetch_mode) && passes(f, i, ws, we, flt))) continue;
        if (k < cap) {
            if (start_out) start_out[k] = f->start[i];
            if (end_out) end_out[k] = f->end[i];
            if (mapq_out) mapq_out[k] = f->mapq[i];
            if (strand_out) strand_out[k] = f->strand ? f->strand[i] : 0;
        }
        ++k;
    }
    return k;
}

/* genome/gaps.py:217-237 (ContigGaps.in_tcmere), all() over telomeres kept */
static int in_tcmere(const orc_gaps* g, int32_t start, int32_t stop) {
    int in_cen = (stop > g->cen_start) && (start < g->cen_stop);
    int in_tel;
    if (g->n_telo == 0) {
        in_tel = 0;
    } else {
        in_tel = 1;
        for (int t = 0; t < g->n_telo; ++t)
            if (!(stop > g->telo_start[t] && start < g->telo_stop[t])) in_tel = 0;
    }
    return in_cen || in_tel;
}

/* frag/_delfi.py:404-472 (_delfi_single_window counts), per window.
 * bl_start/bl_end: the contig's blacklist sorted by (start, stop)
 * (frag/_delfi.py:85-107).  The NOARM gate (:423-428) is the caller's. */
void orc_delfi_counts(const orc_frags* f, const int32_t* ws, const int32_t* we, int64_t n_win, int32_t mapq_min,
                      const int32_t* bl_start, const int32_t* bl_end, int64_t n_bl, const orc_gaps* g,
                      int64_t* short_out, int64_t* long_out, int64_t* nfrag_out) {
    int32_t* reg = (int32_t*)malloc(sizeof(int32_t) * 2 * (size_t)(n_bl > 0 ? n_bl : 1));
    int fetch_mode = f->r1s ? 1 : 0;
    for (int64_t w = 0; w < n_win; ++w) {
        int32_t window_start = ws[w], window_stop = we[w];
        int64_t n_reg = 0;
        /* _blacklist_in_window (frag/_delfi.py:110-126) */
        if (n_bl > 0) {
            int64_t lo = lower_bound_i32(bl_start, n_bl, window_start);
            for (int64_t j = lo; j < n_bl; ++j)
                if (bl_end[j] <= window_stop) { reg[2 * n_reg] = bl_start[j]; reg[2 * n_reg + 1] = bl_end[j]; ++n_reg; }
        }
        int64_t short_lengths = 0, long_lengths = 0, num_frags = 0, lo, hi;
        fetch_range(f, window_start, window_stop, &lo, &hi);
        if (window_stop < window_start) hi = lo;
        for (int64_t i = lo; i < hi; ++i) {
            if (!fetched(f, i, window_start, window_stop, fetch_mode)) continue;
            if ((int32_t)f->mapq[i] < mapq_min) continue;                    /* io/alignment.py:291 */
            int32_t frag_start = f->start[i], frag_stop = f->end[i];
            int32_t frag_length = frag_stop - frag_start;
            if (frag_length < 100 || frag_length > 220) continue;            /* :448 */
            int64_t midpoint = floor_half((int64_t)frag_start + frag_stop);   /* :451 */
            if (midpoint < window_start || midpoint >= window_stop) continue; /* :452 */
            int blacklisted = 0;                                              /* :455-462 */
            for (int64_t r = 0; r < n_reg; ++r) {
                int32_t r0 = reg[2 * r], r1 = reg[2 * r + 1];
                if ((frag_start >= r0 && frag_start < r1) && (frag_stop >= r0 && frag_stop < r1)) { blacklisted = 1; break; }
            }
            if (g && g->has_gaps && in_tcmere(g, frag_start, frag_stop)) continue; /* :464 */
            if (!blacklisted) {                                               /* :467-472 */
                if (frag_length >= 151) long_lengths += 1; else short_lengths += 1;
                num_frags += 1;
            }
        }
        short_out[w] = short_lengths;
        long_out[w] = long_lengths;
        if (nfrag_out) nfrag_out[w] = num_frags;
    }
    free(reg);
}

/* frag/_wps.py:56-205 for one interval [start, stop):
 *   fetch pad (:156-157), frag_array with the midpoint policy (:159-169),
 *   rint windows (:176-178), then _single_nt_wps (:25-53) per base -- kept
 *   O(positions x fragments) like the reference.
 * Returns 0, or -1 if memory ran out. */
int orc_wps(const orc_frags* f, int64_t start, int64_t stop, int64_t chrom_size, int32_t window_size,
            int32_t min_len, int32_t max_len, int32_t mapq_min, int64_t* wps_out) {
    if (stop <= start) return 0;                                  /* :145-152 */
    int64_t minimum = start - max_len; if (minimum < 0) minimum = 0;          /* :156 */
    int64_t maximum = stop + max_len; if (maximum > chrom_size) maximum = chrom_size; /* :157 */
    /* frag_array(start=minimum, stop=maximum, midpoint) */
    int64_t lo = lower_bound_i32(f->start, f->n, minimum - f->max_len);
    int64_t hi = lower_bound_i32(f->start, f->n, maximum + f->hi_slack);
    if (hi < lo) hi = lo;
    int64_t cap = hi - lo, m = 0;
    double* fs = (double*)malloc(sizeof(double) * (size_t)(cap > 0 ? cap : 1));
    double* fe = (double*)malloc(sizeof(double) * (size_t)(cap > 0 ? cap : 1));
    if (!fs || !fe) { free(fs); free(fe); return -1; }
    for (int64_t i = lo; i < hi; ++i) {
        int32_t s = f->start[i], e = f->end[i];
        if (f->r1s) {                                                         /* io/alignment.py:245 */
            if (!((int64_t)f->r1s[i] < maximum && (int64_t)f->r1e[i] > minimum)) continue;
        } else if (!((int64_t)s < maximum && (int64_t)e > minimum)) {         /* io/alignment.py:273-279 */
            continue;
        }
        if ((int32_t)f->mapq[i] < mapq_min) continue;
        int32_t len = e - s;
        if (!(len >= min_len) || !(len <= max_len)) continue;
        int64_t mid = floor_half((int64_t)s + e);
        if (!(mid >= minimum && mid < maximum)) continue;
        fs[m] = (double)s; fe[m] = (double)e; ++m;
    }
    for (int64_t c = start; c < stop; ++c) {
        double window_start = nearbyint((double)c - window_size * 0.5);       /* :177 np.rint */
        double window_stop = nearbyint((double)c + window_size * 0.5 - 1.0);  /* :178, inclusive */
        int64_t num_spanning = 0, num_end_in = 0;
        for (int64_t i = 0; i < m; ++i) {
            int is_spanning = (fs[i] < window_start) && (fe[i] > window_stop);          /* :39-42 */
            int is_start_in = (fs[i] >= window_start) && (fs[i] <= window_stop);        /* :44-46 */
            int is_stop_in = (fe[i] >= window_start) && (fe[i] <= window_stop);         /* :47-49 */
            num_spanning += is_spanning;
            num_end_in += (is_start_in || is_stop_in);                                  /* :50-51 */
        }
        wps_out[c - start] = num_spanning - num_end_in;                                 /* :53 */
    }
    free(fs); free(fe);
    return 0;
}

/* frag/_cleavage_profile.py:33-90 (_coverage_and_ends) over the fragments
 * frag_array(start=adj_start, stop=adj_stop, intersect_policy="any") returns
 * (:204-213): per base depth and fragment-end counts.  min_len / max_len -1 = None. */
void orc_cleavage(const orc_frags* f, int64_t adj_start, int64_t adj_stop, int32_t min_len, int32_t max_len,
                  int32_t mapq_min, int64_t* depth_out, int64_t* ends_out) {
    int64_t n = adj_stop - adj_start;
    if (n <= 0) return;
    int64_t* diff = (int64_t*)calloc((size_t)n + 1, sizeof(int64_t));
    memset(ends_out, 0, (size_t)n * sizeof(int64_t));
    orc_filter flt = {mapq_min, min_len, max_len, 1, f->r1s ? 1 : 0};
    int64_t lo, hi;
    fetch_range(f, (int32_t)adj_start, (int32_t)adj_stop, &lo, &hi);
    for (int64_t i = lo; i < hi; ++i) {
        if (!(fetched(f, i, (int32_t)adj_start, (int32_t)adj_stop, flt.fetch_mode) &&
              passes(f, i, (int32_t)adj_start, (int32_t)adj_stop, &flt)))
            continue;
        int64_t raw_start_idx = (int64_t)f->start[i] - adj_start, raw_stop_idx = (int64_t)f->end[i] - adj_start;
        int64_t a = raw_start_idx < 0 ? 0 : (raw_start_idx > n ? n : raw_start_idx);   /* np.clip(.., 0, n) */
        int64_t b = raw_stop_idx < 0 ? 0 : (raw_stop_idx > n ? n : raw_stop_idx);
        diff[a] += 1;                                                                   /* :72-73 */
        diff[b] -= 1;
        int64_t e = f->strand[i] ? raw_start_idx : raw_stop_idx;                        /* :80-86 */
        if (e >= 0 && e < n) ends_out[e] += 1;
    }
    int64_t run = 0;
    for (int64_t k = 0; k < n; ++k) { run += diff[k]; depth_out[k] = run; }             /* cumsum(diff[:-1]) */
    free(diff);
}

/* ---- all host cores: the per-window work of the functions above on a pthread pool -----------
 * Timed CPU baseline only (bench.py cpu_baseline.all_cores): every task is one window -- coverage
 * count, length histogram, DELFI counts and WPS in 5 kb tiles, exactly the single-thread calls --
 * tasks are handed out through an atomic counter.  Results are thrown away (the single-thread run
 * is the one checked against the GPU); returns the elapsed seconds, or -1. */
#include <pthread.h>
#include <time.h>

typedef struct {
    const orc_frags* f;
    const int32_t *ws, *we;
    int64_t n_win;
    const orc_filter* flt;
    int32_t n_bins, delfi_q;
    const int32_t *bl_start, *bl_end;
    int64_t n_bl;
    const orc_gaps* g;
    int64_t chrom_size;
    int32_t wps_w, wps_min, wps_max, wps_q;
    int64_t next;
    int failed;
} orc_job;

static void* orc_worker(void* arg) {
    orc_job* j = (orc_job*)arg;
    uint32_t* hist = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)j->n_bins);
    int64_t* wps = (int64_t*)malloc(sizeof(int64_t) * 5000);
    if (!hist || !wps) { j->failed = 1; free(hist); free(wps); return NULL; }
    for (;;) {
        int64_t w = __atomic_fetch_add(&j->next, 1, __ATOMIC_RELAXED);
        if (w >= j->n_win) break;
        int64_t cov, over, sh, lg, nf;
        orc_window_counts(j->f, j->ws + w, j->we + w, 1, j->flt, &cov);
        orc_fraglen_hist(j->f, j->ws + w, j->we + w, 1, j->flt, 0, j->n_bins, hist, &over);
        orc_delfi_counts(j->f, j->ws + w, j->we + w, 1, j->delfi_q, j->bl_start, j->bl_end, j->n_bl, j->g, &sh, &lg, &nf);
        for (int64_t x = j->ws[w]; x < j->we[w]; x += 5000) {
            int64_t y = x + 5000 < j->we[w] ? x + 5000 : j->we[w];
            if (orc_wps(j->f, x, y, j->chrom_size, j->wps_w, j->wps_min, j->wps_max, j->wps_q, wps) != 0) j->failed = 1;
        }
    }
    free(hist);
    free(wps);
    return NULL;
}

double orc_all_cores(const orc_frags* f, const int32_t* ws, const int32_t* we, int64_t n_win, const orc_filter* flt,
                     int32_t n_bins, int32_t delfi_q, const int32_t* bl_start, const int32_t* bl_end, int64_t n_bl,
                     const orc_gaps* g, int64_t chrom_size, int32_t wps_w, int32_t wps_min, int32_t wps_max,
                     int32_t wps_q, int32_t n_threads) {
    if (n_threads < 1) n_threads = 1;
    orc_job job = {f, ws, we, n_win, flt, n_bins, delfi_q, bl_start, bl_end, n_bl, g, chrom_size,
                   wps_w, wps_min, wps_max, wps_q, 0, 0};
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * (size_t)n_threads);
    if (!th) return -1.0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int started = 0;
    for (int i = 0; i < n_threads; ++i)
        if (pthread_create(&th[i], NULL, orc_worker, &job) == 0) ++started; else break;
    if (started == 0) orc_worker(&job);
    for (int i = 0; i < started; ++i) pthread_join(th[i], NULL);
    clock_gettime(CLOCK_MONOTONIC, &t1);
    free(th);
    if (job.failed) return -1.0;
    return (double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec);
}
