"""
TEST INFRASTRUCTURE ONLY -- parity checker and CPU baseline, never the product.

Two restatements of the reference's per-window hot path:

* ``C*`` functions: ctypes over ``oracle/libftk_oracle.so`` (``ftk_oracle.c``),
  fast enough for parity at 10^5-10^6 fragments and for the timed
  ``cpu_baseline`` (kind "port") of bench.py.
* ``py_*`` functions: pure-Python loops that follow the reference
  statement-for-statement (per-window fetch, per-fragment predicate); used on
  small cases and as the "reference-shaped single-thread" baseline of
  BASELINE.md section 3.

Both are pinned to the reference by tests/test_oracle_golden.py against the
vectors oracle/gen_golden.py produced by importing the reference.  Only
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

OPEN_LO = -(2 ** 31)
OPEN_HI = 2 ** 31 - 1


class _Filter(C.Structure):
    _fields_ = [("mapq_min", C.c_int32), ("min_len", C.c_int32), ("max_len", C.c_int32),
                ("policy", C.c_int32), ("fetch_mode", C.c_int32)]


class _Gaps(C.Structure):
    _fields_ = [("has_gaps", C.c_int32), ("cen_start", C.c_int32), ("cen_stop", C.c_int32),
                ("n_telo", C.c_int32), ("telo_start", C.c_int32 * 8), ("telo_stop", C.c_int32 * 8)]


class _Frags(C.Structure):
    _fields_ = [("start", C.c_void_p), ("end", C.c_void_p), ("mapq", C.c_void_p), ("strand", C.c_void_p),
                ("r1s", C.c_void_p), ("r1e", C.c_void_p), ("n", C.c_int64), ("max_len", C.c_int32),
                ("hi_slack", C.c_int32)]


def build():
    """Compile the C restatement (gcc) if the shared object is missing/stale."""
    src = os.path.join(_HERE, "ftk_oracle.c")
    so = os.path.join(_HERE, "libftk_oracle.so")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "libftk_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.orc_frag_select.restype = C.c_int64
        _LIB.orc_wps.restype = C.c_int
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Frags:
    """One contig's fragments (file order, start-sorted) for the C oracle."""

    def __init__(self, start, end, mapq, strand=None, r1_start=None, r1_end=None):
        self.start = np.ascontiguousarray(start, dtype=np.int32)
        self.end = np.ascontiguousarray(end, dtype=np.int32)
        self.mapq = np.ascontiguousarray(mapq, dtype=np.uint8)
        self.strand = (np.ascontiguousarray(strand, dtype=np.uint8) if strand is not None
                       else np.zeros(len(self.start), np.uint8))
        self.r1s = None if r1_start is None else np.ascontiguousarray(r1_start, dtype=np.int32)
        self.r1e = None if r1_end is None else np.ascontiguousarray(r1_end, dtype=np.int32)
        self.c = _Frags()
        _lib().orc_frags_init(C.byref(self.c), _p(self.start), _p(self.end), _p(self.mapq), _p(self.strand),
                              _p(self.r1s), _p(self.r1e), C.c_int64(len(self.start)))

    def __len__(self):
        return len(self.start)


def _filter(mapq_min=30, min_len=None, max_len=None, policy="midpoint", bam=False):
    return _Filter(int(mapq_min), -1 if min_len is None else int(min_len), -1 if max_len is None else int(max_len),
                   {"midpoint": 0, "any": 1, "fetch": 2}[policy], 1 if bam else 0)


def _windows(ws, we):
    ws = np.ascontiguousarray([OPEN_LO if v is None else v for v in ws], dtype=np.int32) \
        if not isinstance(ws, np.ndarray) else np.ascontiguousarray(ws, dtype=np.int32)
    we = np.ascontiguousarray([OPEN_HI if v is None else v for v in we], dtype=np.int32) \
        if not isinstance(we, np.ndarray) else np.ascontiguousarray(we, dtype=np.int32)
    return ws, we


def c_window_counts(fr: Frags, ws, we, **flt):
    ws, we = _windows(ws, we)
    out = np.zeros(len(ws), np.int64)
    f = _filter(bam=fr.r1s is not None, **flt)
    _lib().orc_window_counts(C.byref(fr.c), _p(ws), _p(we), C.c_int64(len(ws)), C.byref(f), _p(out))
    return out


def c_fraglen_hist(fr: Frags, ws, we, len_lo, n_bins, **flt):
    ws, we = _windows(ws, we)
    hist = np.zeros((len(ws), n_bins), np.uint32)
    over = np.zeros(len(ws), np.int64)
    f = _filter(bam=fr.r1s is not None, **flt)
    _lib().orc_fraglen_hist(C.byref(fr.c), _p(ws), _p(we), C.c_int64(len(ws)), C.byref(f), C.c_int32(len_lo),
                            C.c_int32(n_bins), _p(hist), _p(over))
    return hist, over


def c_frag_select(fr: Frags, ws, we, **flt):
    ws = OPEN_LO if ws is None else ws
    we = OPEN_HI if we is None else we
    n = len(fr)
    s = np.zeros(n, np.int32); e = np.zeros(n, np.int32); q = np.zeros(n, np.uint8); st = np.zeros(n, np.uint8)
    f = _filter(bam=fr.r1s is not None, **flt)
    k = _lib().orc_frag_select(C.byref(fr.c), C.c_int32(ws), C.c_int32(we), C.byref(f), _p(s), _p(e), _p(q), _p(st),
                               C.c_int64(n))
    return s[:k], e[:k], q[:k], st[:k]


def make_gaps(gaps):
    """gaps: None or (cen_start, cen_stop, [(t0, t1), ...])."""
    g = _Gaps()
    if gaps is None:
        g.has_gaps = 0
        return g
    g.has_gaps = 1
    g.cen_start, g.cen_stop = int(gaps[0]), int(gaps[1])
    tel = list(gaps[2])
    g.n_telo = len(tel)
    for i, (a, b) in enumerate(tel):
        g.telo_start[i] = int(a)
        g.telo_stop[i] = int(b)
    return g


def c_delfi_counts(fr: Frags, ws, we, mapq_min=30, bl_start=None, bl_end=None, gaps=None):
    ws, we = _windows(ws, we)
    n_bl = 0 if bl_start is None else len(bl_start)
    bs = None if n_bl == 0 else np.ascontiguousarray(bl_start, dtype=np.int32)
    be = None if n_bl == 0 else np.ascontiguousarray(bl_end, dtype=np.int32)
    sh = np.zeros(len(ws), np.int64); lg = np.zeros(len(ws), np.int64); nf = np.zeros(len(ws), np.int64)
    g = make_gaps(gaps)
    _lib().orc_delfi_counts(C.byref(fr.c), _p(ws), _p(we), C.c_int64(len(ws)), C.c_int32(mapq_min), _p(bs), _p(be),
                            C.c_int64(n_bl), C.byref(g), _p(sh), _p(lg), _p(nf))
    return sh, lg, nf


def c_all_cores(fr: Frags, ws, we, n_bins, mapq_min, bl_start, bl_end, gaps, chrom_size, wps_w, wps_min, wps_max,
                n_threads):
    """Seconds the per-window work (counts + histogram + DELFI + WPS in 5 kb tiles) of ``len(ws)`` windows takes
    on ``n_threads`` pthreads (timed CPU baseline; results are not returned)."""
    ws, we = _windows(ws, we)
    n_bl = 0 if bl_start is None else len(bl_start)
    bs = None if n_bl == 0 else np.ascontiguousarray(bl_start, dtype=np.int32)
    be = None if n_bl == 0 else np.ascontiguousarray(bl_end, dtype=np.int32)
    f = _filter(mapq_min=mapq_min, bam=fr.r1s is not None)
    g = make_gaps(gaps)
    fn = _lib().orc_all_cores
    fn.restype = C.c_double
    return float(fn(C.byref(fr.c), _p(ws), _p(we), C.c_int64(len(ws)), C.byref(f), C.c_int32(n_bins), C.c_int32(mapq_min),
                    _p(bs), _p(be), C.c_int64(n_bl), C.byref(g), C.c_int64(chrom_size), C.c_int32(wps_w),
                    C.c_int32(wps_min), C.c_int32(wps_max), C.c_int32(mapq_min), C.c_int32(n_threads)))


def c_wps(fr: Frags, start, stop, chrom_size, window_size=120, min_len=120, max_len=180, mapq_min=30):
    out = np.zeros(max(int(stop) - int(start), 0), np.int64)
    rc = _lib().orc_wps(C.byref(fr.c), C.c_int64(start), C.c_int64(stop), C.c_int64(chrom_size),
                        C.c_int32(window_size), C.c_int32(min_len), C.c_int32(max_len), C.c_int32(mapq_min), _p(out))
    if rc != 0:
        raise MemoryError("orc_wps")
    return out


def c_cleavage(fr: Frags, adj_start, adj_stop, min_len=None, max_len=None, mapq_min=30):
    """(depth, ends, proportion) per base; proportion as frag/_cleavage_profile.py:208-210."""
    n = max(int(adj_stop) - int(adj_start), 0)
    depth = np.zeros(n, np.int64)
    ends = np.zeros(n, np.int64)
    _lib().orc_cleavage(C.byref(fr.c), C.c_int64(adj_start), C.c_int64(adj_stop),
                        C.c_int32(-1 if min_len is None else min_len), C.c_int32(-1 if max_len is None else max_len),
                        C.c_int32(mapq_min), _p(depth), _p(ends))
    prop = np.zeros(n, np.float64)
    nz = depth != 0
    prop[nz] = ends[nz] / depth[nz] * 100
    return depth, ends, prop


# ---------------------------------------------------------------------------
# BAM records -> rows (restates io/alignment.py:60-71,242-268 over records read with gzip + struct; pinned to the
# reference's own code by tests/golden/bam.json.gz, oracle/gen_golden_bam.py)
# ---------------------------------------------------------------------------
def bam_rows(path, read1_only=True):
    """``(names, lengths, {contig: rows})``: per contig the rows ``(fs, fe, mapq, fwd, r1s, r1e)`` the reference's
    ``_fetch_sam`` yields for a whole-contig fetch at ``quality_threshold=0``, in FILE order; ``[r1s, r1e)`` is the
    alignment htslib's region iterator tests (``bam_endpos``: a reference length of 0 counts as 1).  Raises TypeError
    where the reference does (a CIGAR-less read1 with TLEN < 0: ``None + tlen``, :257)."""
    import gzip
    import struct
    with gzip.open(path, "rb") as fh:
        data = fh.read()
    assert data[:4] == b"BAM\1"
    (l_text,) = struct.unpack_from("<i", data, 4)
    o = 8 + l_text
    (n_ref,) = struct.unpack_from("<i", data, o)
    o += 4
    names, lengths = [], []
    for _ in range(n_ref):
        (l_name,) = struct.unpack_from("<i", data, o)
        names.append(data[o + 4:o + 4 + l_name - 1].decode())
        lengths.append(struct.unpack_from("<i", data, o + 4 + l_name)[0])
        o += 8 + l_name
    out = {n: [] for n in names}
    while o + 4 <= len(data):
        (bs,) = struct.unpack_from("<i", data, o)
        ref_id, pos, l_name, mapq, _bin, n_cigar, flag, _l_seq, _nref, _npos, tlen = struct.unpack_from("<iiBBHHHiiii", data, o + 4)
        cigar = struct.unpack_from(f"<{n_cigar}I", data, o + 4 + 32 + l_name)
        o += 4 + bs
        if ref_id < 0:
            continue
        # _read_is_low_quality (:60-71) at quality_threshold 0
        if (flag & 0x4) or (flag & 0x100) or not (flag & 0x1) or (flag & 0x8) or (flag & 0x400) or (flag & 0x200) \
                or (flag & 0x800) or not (flag & 0x2):
            continue
        if read1_only and (flag & 0x80):  # :248
            continue
        rlen = sum(v >> 4 for v in cigar if (v & 15) in (0, 2, 3, 7, 8))
        end_pos = pos + (rlen or 1)                       # htslib bam_endpos
        reference_end = end_pos if n_cigar else None      # pysam: None without a CIGAR
        if tlen > 0:                                      # :253-255
            fs, fe = pos, pos + tlen
        elif tlen < 0:                                    # :256-258
            fs, fe = reference_end + tlen, reference_end
        else:
            continue
        out[names[ref_id]].append((fs, fe, mapq, 0 if flag & 0x10 else 1, pos, end_pos))
    return names, lengths, out


def frags_from_bam_rows(rows):
    """``Frags`` (start-sorted, stable: equal starts keep file order) of one contig's ``bam_rows`` + the file rank
    of every sorted row."""
    order = sorted(range(len(rows)), key=lambda j: rows[j][0])
    a = np.array([rows[j] for j in order], dtype=np.int64).reshape(-1, 6)
    return Frags(a[:, 0], a[:, 1], a[:, 2], a[:, 3], a[:, 4], a[:, 5]), np.array(order, np.int64)


# ---------------------------------------------------------------------------
# Pure-Python, reference-shaped restatement (small cases / timed baseline)
# ---------------------------------------------------------------------------
def _none_geq(a, b):  # utils/_comparison.py:20-24
    return True if a is None or b is None else a >= b


def _none_leq(a, b):  # utils/_comparison.py:13-17
    return True if a is None or b is None else a <= b


def py_fetch(rows, start, stop, quality_threshold):
    """io/alignment.py:270-302 over in-memory rows ``(fs, fe, mapq, fwd)``
    of one contig: tabix overlap query, then the mapq cut.  Rows of a BAM
    (``bam_rows``) carry two more fields, read1's alignment ``[r1s, r1e)``:
    that is what the index query tests there (io/alignment.py:245)."""
    for row in rows:
        fs, fe, mapq, fwd = row[:4]
        a, b = (row[4], row[5]) if len(row) > 4 else (fs, fe)
        if stop is not None and not a < stop:
            continue
        if start is not None and not b > start:
            continue
        if mapq < quality_threshold:
            continue
        yield fs, fe, mapq, fwd


def py_frag_generator(rows, start, stop, min_length, max_length, intersect_policy, quality_threshold):
    """utils/_frag_generator.py:58-141 (stream of passing fragments)."""
    for fs, fe, mapq, fwd in py_fetch(rows, start, stop, quality_threshold):
        length = fe - fs
        if not (_none_geq(length, min_length) and _none_leq(length, max_length)):
            continue
        if intersect_policy == "midpoint":  # :35-42
            midpoint = (fs + fe) // 2
            ok = (start is None or midpoint >= start) and (stop is None or midpoint < stop)
        elif intersect_policy == "any":  # :44-50
            ok = (start is None or fe > start) and (stop is None or fs < stop)
        else:
            raise ValueError(intersect_policy)
        if ok:
            yield fs, fe, mapq, fwd


def py_single_coverage(rows, start=0, stop=None, min_length=None, max_length=None, intersect_policy="midpoint",
                       quality_threshold=30):
    """frag/_coverage.py:117-130."""
    coverage = 0
    for _ in py_frag_generator(rows, start, stop, min_length, max_length, intersect_policy, quality_threshold):
        coverage += 1
    return coverage


def py_distribution(rows, start, stop, min_length, max_length, intersect_policy, quality_threshold):
    """frag/_frag_length.py:147-153."""
    value_counts = {}
    for fs, fe, _, _ in py_frag_generator(rows, start, stop, min_length, max_length, intersect_policy,
                                          quality_threshold):
        length = fe - fs
        value_counts[length] = value_counts.get(length, 0) + 1
    return value_counts


def py_find_median(val_freq_dict):
    """frag/_frag_length.py:156-172, including the odd-count search for
    ``total // 2`` (not ``total // 2 + 1``)."""
    val = np.array(list(val_freq_dict.keys()))
    freq = np.array(list(val_freq_dict.values()))
    order = np.argsort(val)
    val = val[order]
    freq = freq[order]
    cdf = np.cumsum(freq)
    total_count = cdf[-1]
    if total_count % 2 == 1:
        return float(val[np.searchsorted(cdf, total_count // 2)])
    idx = np.searchsorted(cdf, [total_count // 2, total_count // 2 + 1])
    return float(np.mean(val[idx]))


def py_frag_length_stats(dist, short_reads=150):
    """frag/_frag_length.py:202-238 -> (mean, median, stdev, min, max, n, frac_short)."""
    total_count = sum(dist.values())
    if total_count == 0:
        return (-1, -1, -1, -1, -1, -1, -1)
    mean = sum(v * c for v, c in dist.items()) / total_count
    median = py_find_median(dist)
    variance = sum(c * ((v - mean) ** 2) for v, c in dist.items()) / total_count
    stdev = variance ** 0.5
    n_short = sum(c for v, c in dist.items() if v <= short_reads)
    return (mean, median, stdev, min(dist.keys()), max(dist.keys()), total_count, n_short / total_count)


def py_delfi_single_window(rows, window_start, window_stop, quality_threshold, blacklist, gaps):
    """frag/_delfi.py:404-472.  blacklist: sorted list of (r0, r1) for the
    contig; gaps: None or (cen_start, cen_stop, [(t0, t1), ...])."""
    def in_tcmere(start, stop):  # genome/gaps.py:217-237
        in_cen = stop > gaps[0] and start < gaps[1]
        in_tel = all(stop > t[0] and start < t[1] for t in gaps[2]) if gaps[2] else False
        return in_cen or in_tel

    regions = [r for r in blacklist if r[0] >= window_start and r[1] <= window_stop]  # :110-126
    short_lengths = long_lengths = num_frags = 0
    for fs, fe, _, _ in py_fetch(rows, window_start, window_stop, quality_threshold):
        frag_length = fe - fs
        if frag_length < 100 or frag_length > 220:
            continue
        midpoint = (fs + fe) // 2
        if midpoint < window_start or midpoint >= window_stop:
            continue
        blacklisted = False
        for r in regions:
            if (fs >= r[0] and fs < r[1]) and (fe >= r[0] and fe < r[1]):
                blacklisted = True
                break
        if gaps is not None and in_tcmere(fs, fe):
            continue
        if not blacklisted:
            if frag_length >= 151:
                long_lengths += 1
            else:
                short_lengths += 1
            num_frags += 1
    return short_lengths, long_lengths, num_frags


def py_wps(rows, start, stop, chrom_size, window_size=120, min_length=120, max_length=180, quality_threshold=30):
    """frag/_wps.py:156-188 with the numpy form of _single_nt_wps (:25-53)."""
    if stop <= start:
        return np.zeros(0, np.int64)
    minimum = max(round(start - max_length), 0)
    maximum = min(round(stop + max_length), chrom_size)
    sel = [(fs, fe) for fs, fe, _, _ in py_frag_generator(rows, minimum, maximum, min_length, max_length,
                                                          "midpoint", quality_threshold)]
    fs = np.array([s for s, _ in sel], dtype=np.int64)
    fe = np.array([e for _, e in sel], dtype=np.int64)
    centers = np.arange(start, stop, dtype=np.int64)
    w0 = np.rint(centers - window_size * 0.5)
    w1 = np.rint(centers + window_size * 0.5 - 1)
    out = np.zeros(stop - start, np.int64)
    for i in range(stop - start):
        spanning = np.sum((fs < w0[i]) * (fe > w1[i]))
        start_in = (fs >= w0[i]) * (fs <= w1[i])
        stop_in = (fe >= w0[i]) * (fe <= w1[i])
        out[i] = spanning - np.sum(np.logical_or(start_in, stop_in))
    return out


# ---------------------------------------------------------------------------
# adjust_wps (frag/_adjust_wps.py:25-50,119-140), one run of consecutive scores
# ---------------------------------------------------------------------------
def py_adjust_run(scores, window_size=1000, use_mean=False, edge_size=None, savgol_window=21, savgol_deg=2,
                  savgol=True):
    """``edge_size`` not None = subtract_edges.  The running statistic is stated with an explicit sort
    of every window (middle pair averaged) instead of np.median; the Savitzky-Golay pass is scipy's,
    as in the reference."""
    x = np.asarray(scores, np.float64)
    if edge_size is not None:
        x = x - np.mean([np.mean(x[:edge_size]), np.mean(x[-edge_size:])])
    if window_size > len(x):
        raise ValueError("median_window_size cannot be greater than the length of interval")
    n = len(x) - window_size
    h = window_size // 2
    running = np.empty(max(n, 0), np.float64)
    for i in range(n):
        w = x[i:i + window_size]
        if use_mean:
            running[i] = np.mean(w)
        else:
            s = np.sort(w)
            running[i] = (s[h - 1] + s[h]) / 2.0 if window_size % 2 == 0 else s[h]
    adj = x[h:len(x) - h] - running
    if savgol:
        from scipy.signal import savgol_filter
        adj = savgol_filter(adj, savgol_window, savgol_deg)
    return adj


# ---------------------------------------------------------------------------
# end / breakpoint motifs (frag/_end_motifs.py:118-176, frag/_breakpoint_motifs.py:124-185)
# ---------------------------------------------------------------------------
_COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def _py_sequence(seq, start, stop):
    """io/reference.py:120-172: upper-cased slice, ValueError outside the contig."""
    if seq is None or start < 0 or stop > len(seq) or start > stop:
        raise ValueError("out of bounds")
    return seq[start:stop].upper()


def py_region_motifs(rows, seq, start, stop, k, kind="end", both_strands=True, negative_strand=False,
                     quality_threshold=20):
    """Counts per k-mer in gen_kmers order for one region.  ``rows``: the contig's
    ``(fs, fe, mapq, fwd)`` in file order; ``seq``: the contig's reference sequence (None = contig
    absent from the genome).  kind "end": k-mer at [fs, fs+k) and revcomp of [fe-k, fe);
    kind "breakpoint": [fs-k//2, fs+k//2) and revcomp of [fe-k//2, fe+k//2) with the contig-end guard.
    Raises RuntimeError where the reference does (end motifs, both strands, 3' k-mer off the contig)."""
    import itertools
    if both_strands and negative_strand:
        raise ValueError("Cannot have both both_strands and negative_strand.")
    kmers = ["".join(t) for t in itertools.product("ACGT", repeat=k)]
    counts = dict.fromkeys(kmers, 0)
    h = k // 2
    chrom_len = len(seq) if seq is not None else 0
    for fs, fe, _, fwd in py_fetch(rows, start, stop, quality_threshold):
        if kind == "breakpoint":
            if fs - h < 0 or fs + h >= chrom_len:
                continue
            use_fwd = both_strands or (fwd and not negative_strand)
            use_rev = both_strands or negative_strand
            f_lo, f_hi, r_lo, r_hi = fs - h, fs + h, fe - h, fe + h
        else:
            use_fwd = both_strands or (fwd and not negative_strand)
            use_rev = both_strands or negative_strand
            f_lo, f_hi, r_lo, r_hi = fs, fs + k, fe - k, fe
        if use_fwd:
            try:
                kmer = _py_sequence(seq, f_lo, f_hi)
            except ValueError:
                continue
            if len(kmer) != k:
                continue
            if "N" not in kmer:
                counts[kmer] += 1
        if use_rev:
            try:
                kmer = _py_sequence(seq, r_lo, r_hi)
            except ValueError:
                if kind == "end" and both_strands:
                    raise RuntimeError("Error querying sequence")
                continue
            if len(kmer) != k:
                continue
            if "N" not in kmer:
                counts["".join(_COMP[b] for b in reversed(kmer))] += 1
    return np.array([counts[m] for m in kmers], np.int64)
