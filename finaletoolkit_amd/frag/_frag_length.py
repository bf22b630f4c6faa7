"""
Fragment-length features with the reference's surface
(``src/finaletoolkit/frag/_frag_length.py``): ``frag_length`` (raw lengths),
``frag_length_bins`` (binned distribution + statistics) and
``frag_length_intervals`` (per-interval statistics).

The ``length -> count`` dictionaries of ``_distribution_from_gen`` (``:147-153``)
come from the ``ftk_fraglen_hist`` kernel; the statistics (``:156-238``) are host
arithmetic on those histograms with the reference's exact formulas, including
the odd-count median search for ``total // 2``.
"""
from __future__ import annotations

import gzip
import queue
import sys
import threading
import time
import warnings
from pathlib import Path
from sys import stderr, stdout
from typing import NamedTuple, Union

import numpy as np

from .. import sharding, writers
from .._stages import Stages, without_collector
from ..source import ContigFeed, get_engine, open_source
from ..utils import _check_policy, _check_region, _region_contigs, get_intervals

__all__ = ["frag_length", "frag_length_bins", "frag_length_intervals", "FragLengthStats"]

_MAX_BINS = 32768               # kHistMaxBins of the kernel
_DEVICE_STATS = __import__("os").environ.get("FTK_DEVICE_STATS", "1") != "0"  # 0: the numpy statistics (tests hold the two together)
_HIST_BYTES_PER_CALL = 1 << 29  # window batches are cut to keep one histogram block <= 512 MiB


LAST_STAGE_S: dict = {}  # the last frag_length_bins / frag_length_intervals call's wall time by stage (seconds)


class FragLengthStats(NamedTuple):
    """Per-interval length statistics (frag/_frag_length.py:32-75); all ``-1`` when empty."""

    contig: str
    start: int
    stop: int
    name: str
    mean: float
    median: float
    stdev: float
    minimum: int
    maximum: int
    count: int
    frac_short_reads: float


def _find_median(values: np.ndarray, freq: np.ndarray) -> float:
    """frag/_frag_length.py:156-172 on sorted ``values`` with counts ``freq``."""
    cdf = np.cumsum(freq)
    total_count = cdf[-1]
    if total_count % 2 == 1:
        return float(values[np.searchsorted(cdf, total_count // 2)])
    idx = np.searchsorted(cdf, [total_count // 2, total_count // 2 + 1])
    return float(np.mean(values[idx]))


def _stats_from_dist(values: np.ndarray, freq: np.ndarray, short_cut):
    """(mean, median, stdev, min, max, total, n_short) from sorted lengths/counts
    (frag/_frag_length.py:202-224, :432-456)."""
    vals = [int(v) for v in values]
    cnts = [int(c) for c in freq]
    total_count = sum(cnts)
    mean = sum(v * c for v, c in zip(vals, cnts)) / total_count
    median = _find_median(values, freq)
    variance = sum(c * ((v - mean) ** 2) for v, c in zip(vals, cnts)) / total_count
    n_short = None if short_cut is None else sum(c for v, c in zip(vals, cnts) if v <= short_cut)
    return mean, median, variance ** 0.5, vals[0], vals[-1], total_count, n_short


def _stats_rows(h: np.ndarray, lo: int, short_cut):
    """``_stats_from_dist`` for every row of a dense histogram block ``h[r, b]`` = count of length ``lo + b``
    at once (numpy over the block instead of a Python loop per interval and per length): arrays
    ``(mean, median, stdev, min, max, total, n_short)``; rows without fragments have total 0 (other entries
    undefined).  Same formulas: the mean from exact integer sums, the median by the reference's search (odd
    totals look for ``total // 2``), the variance around that mean (numpy's summation order instead of
    ascending lengths: ~1e-16 relative)."""
    n_rows, n_b = h.shape
    v = lo + np.arange(n_b, dtype=np.int64)
    tot = h.sum(axis=1)
    safe = np.maximum(tot, 1)
    present = h > 0
    first = present.argmax(axis=1)
    last = n_b - 1 - present[:, ::-1].argmax(axis=1)
    mean = (h @ v) / safe
    var = (((v[None, :] - mean[:, None]) ** 2) * h).sum(axis=1) / safe
    cdf = np.cumsum(h, axis=1)
    k1 = tot // 2
    i1 = np.where(k1 == 0, first, (cdf >= k1[:, None]).argmax(axis=1))  # searchsorted(cdf, 0) is the first value
    i2 = (cdf >= (k1 + 1)[:, None]).argmax(axis=1)
    median = np.where(tot % 2 == 1, v[i1].astype(np.float64), (v[i1] + v[i2]) / 2.0)
    n_short = None if short_cut is None else h[:, v <= short_cut].sum(axis=1)
    return mean, median, np.sqrt(var), v[first], v[last], tot, n_short


def _length_range(eng, key, min_length, max_length):
    _, data_max, _ = eng.info(key)
    lo = 0 if min_length is None else max(int(min_length), 0)
    hi = data_max if max_length is None else min(int(max_length), data_max)
    return lo, hi


def _window_hists(eng, key, ws, we, lo, hi, quality_threshold, min_length, max_length, intersect_policy):
    """Dense histograms [n_win, hi - lo + 1] (uint32->int64), length range split when it exceeds the kernel limit."""
    n_win = len(ws)
    if hi < lo or n_win == 0:
        return np.zeros((n_win, 0), np.int64)
    n_total = hi - lo + 1
    out = np.zeros((n_win, n_total), np.int64)
    for b0 in range(0, n_total, _MAX_BINS):
        nb = min(_MAX_BINS, n_total - b0)
        step = max(1, _HIST_BYTES_PER_CALL // (4 * nb))
        for w0 in range(0, n_win, step):
            h, _ = eng.fraglen_hist(key, ws[w0:w0 + step], we[w0:w0 + step], lo + b0, nb, quality_threshold,
                                    min_length, max_length, intersect_policy)
            out[w0:w0 + step, b0:b0 + nb] = h
    return out


def frag_length(input_file: Union[str, Path], contig: str | None = None, start: int | None = None,
                stop: int | None = None, intersect_policy: str = "midpoint", output_file: str | None = None,
                quality_threshold: int = 30, verbose: bool = False,
                reference_file: str | Path | None = None) -> np.ndarray:
    """``int32`` array of fragment lengths in file order (frag/_frag_length.py:246-330)."""
    if verbose:
        t0 = time.time()
        stderr.write("Finding frag lengths.\n")
    _check_policy(intersect_policy)
    _check_region(contig, start, stop)
    src = open_source(input_file)
    src.check_fetch(contig, start, stop)
    eng = get_engine()
    names, whole = _region_contigs(src, contig)
    parts = [eng.frag_lengths(src.require(c) if whole else src.require_interval(c, start, stop, 1), None if whole else start, None if whole else stop, quality_threshold,
                              0, 1000000000, intersect_policy) for c in names]
    lengths = np.concatenate(parts).astype(np.int32) if parts else np.zeros(0, np.int32)

    if isinstance(output_file, str):
        if not (output_file.endswith(".bin") or output_file == "-"):
            raise ValueError("output_file can only have suffixes .wig or .wig.gz.")
        if not sharding.is_writer():
            pass  # a region query is not sharded: every rank holds the result, rank 0 alone writes it
        elif output_file.endswith(".bin"):
            with open(output_file, "wb") as out:
                lengths.tofile(out)
        else:
            for line in lengths:
                stdout.write(f"{line}\n")
    elif output_file is not None:
        raise TypeError(f'output_file is unsupported type "{type(input_file)}". output_file should be a string '
                        "specifying the path of the file to write output scores to.")
    if verbose:
        stderr.write(f"frag_length took {time.time() - t0} s to complete\n")
    return lengths


@without_collector
def frag_length_bins(input_file, contig: str | None = None, start: int | None = None, stop: int | None = None,
                     min_length: int | None = 0, max_length: int | None = None, bin_size: int = 1,
                     output_file: str | None = None, intersect_policy: str = "midpoint", quality_threshold: int = 30,
                     summary_stats: bool = False, short_fraction: int | None = None,
                     histogram_path: str | None = None, verbose: Union[bool, int] = False,
                     reference_file: str | Path | None = None):
    """Binned length distribution of a region / contig / the whole file
    (frag/_frag_length.py:333-508).  Returns ``(bins, counts)``."""
    if verbose:
        t0 = time.time()
        stderr.write("Generating fragment dictionary. \n")
    _check_policy(intersect_policy)
    _check_region(contig, start, stop)
    clock = Stages()
    src = open_source(input_file)
    src.check_fetch(contig, start, stop)
    eng = get_engine()
    clock.lap("open")
    if contig is None and sharding.rank_world()[1] > 1:
        # (several ranks: a rank decodes only the contigs it is dealt below - not the whole file, as ``load_all`` would)
        names, whole = [c for c in src.contigs if src.has(c)], True
    else:
        names, whole = _region_contigs(src, contig)
    # the whole file: every fragment counts once, so the contigs are dealt WHOLE to the ranks of the process group (LPT
    # on their lengths; a region read hands a fragment to every rank whose region it overlaps); the sparse length ->
    # count maps meet in one all-gather of small objects; a single contig / region is counted by every rank alike
    clock.lap("decode_wait")
    rank, world, owner = sharding.contig_owner({c: float(src.lengths.get(c) or 1) for c in names})
    shard = world > 1 and len(names) > 1
    dist: dict[int, int] = {}
    err = None
    try:
        for c in names:
            if shard and owner[c] != rank:
                continue
            key = src.require(c)
            clock.lap("decode_wait")
            lo, hi = _length_range(eng, key, min_length, max_length)
            h = _window_hists(eng, key, [None if whole else start], [None if whole else stop], lo, hi,
                              quality_threshold, min_length, max_length, intersect_policy)
            clock.lap("histograms")
            if h.shape[1]:
                for b in np.nonzero(h[0])[0]:
                    dist[lo + int(b)] = dist.get(lo + int(b), 0) + int(h[0, b])
    except Exception as e:  # noqa: BLE001 - every rank learns of it below
        err = e
    if world > 1:
        sharding.agree(err)
    elif err is not None:
        raise err
    if shard:
        merged: dict[int, int] = {}
        for part in sharding.allgather_object(dist):
            for length, count in part.items():
                merged[length] = merged.get(length, 0) + count
        dist = merged
    total_count = sum(dist.values())
    if total_count == 0:
        warnings.warn("No fragments found in the specified region. Returning empty result.", RuntimeWarning,
                      stacklevel=2)
        return np.array([]), np.array([])

    values = np.array(sorted(dist), dtype=np.int64)
    freq = np.array([dist[int(v)] for v in values], dtype=np.int64)
    mean, median, stdev, vmin, vmax, _, n_short = _stats_from_dist(values, freq, short_fraction)
    stats = [("mean", mean), ("median", median), ("stdev", stdev), ("min", vmin), ("max", vmax),
             ("total count", total_count)]
    if short_fraction is not None:
        stats.append((f"short fraction (s{short_fraction})", n_short / total_count))

    bin_start, bin_stop = vmin, vmax
    n_bins = (bin_stop - bin_start) // bin_size
    bins = np.arange(bin_start, bin_stop + bin_size, bin_size)
    counts_arr = np.zeros(n_bins + 1, dtype=np.int64)
    np.add.at(counts_arr, (values - bin_start) // bin_size, freq)
    counts = counts_arr.tolist()

    if output_file is not None and sharding.is_writer():
        writers.write_length_bins(output_file, bins, counts, bin_size, stats if summary_stats else None)
    clock.lap("statistics_and_write")
    clock.publish(LAST_STAGE_S)
    if histogram_path is not None:
        raise NotImplementedError("histogram plotting (matplotlib) is outside the MI355X hot path")
    if verbose:
        stderr.write(f"frag_length_bins took {time.time() - t0} s to complete.\n")
    return bins, counts


def _result_rows(results: list, intervals, index, stats: np.ndarray, lines: list | None = None) -> None:
    """``results[i] = FragLengthStats(...)`` for the intervals ``index`` from their statistics rows ``stats[k]`` =
    mean median stdev min max count n_short; an interval without a fragment: every statistic is the integer -1
    (frag/_frag_length.py:202-238).  ``lines[i]``, when asked for, is the interval's row of the output file (the
    eleven fields, tab-separated, ``str()`` of each) - made here, contig by contig while the decoder works on the next
    one, instead of in one go behind the last contig."""
    import gc
    total = stats[:, 5].astype(np.int64)
    frac = np.divide(stats[:, 6].astype(np.int64), total, out=np.zeros(len(total), np.float64), where=total > 0)  # int / int
    cols = zip(np.asarray(index).tolist(), stats[:, 0].tolist(), stats[:, 1].tolist(), stats[:, 2].tolist(),
               stats[:, 3].astype(np.int64).tolist(), stats[:, 4].astype(np.int64).tolist(), total.tolist(), frac.tolist())
    was = gc.isenabled()
    gc.disable()  # (tens of thousands of small tuples: the collector would walk them again and again)
    try:
        for i, mean, median, stdev, vmin, vmax, n, short in cols:
            c, a, b, name = intervals[i]
            if n:
                results[i] = FragLengthStats(c, a, b, name, mean, median, stdev, vmin, vmax, n, short)
                if lines is not None:
                    lines[i] = f"{c}\t{a}\t{b}\t{name}\t{mean}\t{median}\t{stdev}\t{vmin}\t{vmax}\t{n}\t{short}"
            else:
                results[i] = FragLengthStats(c, a, b, name, -1, -1, -1, -1, -1, -1, -1)
                if lines is not None:
                    lines[i] = f"{c}\t{a}\t{b}\t{name}\t-1\t-1\t-1\t-1\t-1\t-1\t-1"
    finally:
        if was:
            gc.enable()


@without_collector
def frag_length_intervals(input_file, interval_file: str, output_file: str | None = None,
                          min_length: int | None = 0, max_length: int | None = None, quality_threshold: int = 30,
                          intersect_policy: str = "midpoint", short_reads: int = 150, workers: int = 1,
                          verbose: Union[bool, int] = False,
                          reference_file: str | Path | None = None) -> list[FragLengthStats]:
    """Per-interval length statistics over a BED file (frag/_frag_length.py:511-640)."""
    if verbose:
        t0 = time.time()
        stderr.write("Reading intervals.\n")
    _check_policy(intersect_policy)
    clock = Stages()
    one_process = sharding.rank_world()[1] == 1
    eng = get_engine()
    if not one_process:
        src = open_source(input_file, workers)
        clock.lap("open")
    intervals = get_intervals(interval_file)
    clock.lap("read_intervals")
    results: list = [None] * len(intervals)
    lines = [None] * len(intervals) if output_file is not None and sharding.is_writer() else None
    # Pool(workers) of the reference (:571-593) = one rank per GPU: the intervals are cut into equal-cost runs over the
    # ranks (sharding.IntervalPlan: whole contigs, a region of the contig a cut falls into), a rank decodes and counts
    # only its share, and one all-gather of the seven statistics per interval (float64 bit patterns) gives every rank
    # the whole list; rank 0 writes.
    iv_starts = np.array([iv[1] for iv in intervals], dtype=np.int64)
    iv_stops = np.array([iv[2] for iv in intervals], dtype=np.int64)

    def unit_stats(key, idx):
        """float64 [len(idx), 7]: mean median stdev min max total n_short; total 0 = no fragment"""
        out = np.zeros((len(idx), 7), np.float64)
        lo, hi = _length_range(eng, key, min_length, max_length)
        ws = iv_starts[idx].astype(np.int32)
        we = iv_stops[idx].astype(np.int32)
        if lo <= hi and hi - lo + 1 <= _MAX_BINS and _DEVICE_STATS:
            # the statistics on the device, from histogram rows that never leave it (ftk_fraglen_stats)
            step = max(1, _HIST_BYTES_PER_CALL // (4 * (hi - lo + 1)))
            for w0 in range(0, len(idx), step):
                out[w0:w0 + step] = eng.fraglen_stats(key, ws[w0:w0 + step], we[w0:w0 + step], lo, hi - lo + 1, short_reads,
                                                      quality_threshold, min_length, max_length, intersect_policy)
            return out
        n_total = max(hi - lo + 1, 1)
        step = max(1, _HIST_BYTES_PER_CALL // (8 * n_total))
        for w0 in range(0, len(idx), step):
            h = _window_hists(eng, key, ws[w0:w0 + step], we[w0:w0 + step], lo, hi, quality_threshold, min_length,
                              max_length, intersect_policy)
            if h.shape[1] == 0:
                continue
            rows_per = max(1, (1 << 22) // h.shape[1])  # statistics of a few thousand intervals at a time
            for r0 in range(0, h.shape[0], rows_per):
                blk = h[r0:r0 + rows_per]
                cols = _stats_rows(blk, lo, short_reads)
                dst = out[w0 + r0:w0 + r0 + blk.shape[0]]
                for k in range(7):
                    if cols[k] is not None:
                        dst[:, k] = cols[k]
                dst[cols[5] == 0] = 0.0
        return out

    if one_process:
        # One process: the decode runs ahead on a helper thread (source.ContigFeed: only the intervals' contigs are
        # wanted) and a contig's intervals are counted and summarised as soon as it is resident.
        by_contig: dict = {}
        for i, iv in enumerate(intervals):
            by_contig.setdefault(iv[0], []).append(i)
        stats = None
        feed = ContigFeed(input_file, workers, names=list(by_contig))
        # (starting the decode before the intervals are parsed - the contigs named by a quick pass over the BED file - was
        # measured: the parse and the decoder's start slow each other down by what the head start gains)
        # A contig's rows (named tuples, output lines: 1.5 us of interpreter time per interval) are made on a second
        # thread while this one waits - without the interpreter lock - for the next contig or for its kernels: made
        # here, between two contigs, they were longer than the decoder needs for a contig and set the command's pace.
        todo: queue.SimpleQueue = queue.SimpleQueue()
        failed: list = []

        def make_rows():
            while True:
                item = todo.get()
                if item is None:
                    return
                try:
                    _result_rows(results, intervals, item[0], item[1], lines)
                except BaseException as e:  # noqa: BLE001 - re-raised by the command's own thread below
                    failed.append(e)
                    return

        rows_thread = threading.Thread(target=make_rows, name="ftk-length-rows", daemon=True)
        # (the row thread is pure interpreter work: with the default 5 ms switch interval this thread would wait that
        # long for the lock every time a contig arrives or a kernel returns)
        switch = sys.getswitchinterval()
        sys.setswitchinterval(min(switch, 2e-4))
        rows_thread.start()
        try:
            for src, c in feed:
                clock.lap("decode_wait")
                idx = by_contig.pop(c, None)
                if idx is not None:
                    idx = np.asarray(idx, dtype=np.int64)
                    order = idx[np.argsort(iv_starts[idx], kind="stable")]
                    block = unit_stats(src.key(c), order)
                    clock.lap("histograms_and_statistics")
                    todo.put((order, block))
            src = feed.finish()
        except BaseException:
            feed.close()
            raise
        finally:
            todo.put(None)
            rows_thread.join()
            sys.setswitchinterval(switch)
        if failed:
            raise failed[0]
        clock.lap("result_rows_tail")
        for c in by_contig:  # contigs the file does not hold: the error the reference's fetch raises (ValueError)
            src.require(c)
    else:
        plan = sharding.IntervalPlan([iv[0] for iv in intervals], iv_starts.tolist(), iv_stops.tolist())
        local, err = {}, None
        try:
            for unit in plan.mine:
                key = plan.unit_key(src, unit, 1)
                clock.lap("decode_wait")
                try:
                    local[unit] = unit_stats(key, plan.intervals(unit))
                finally:
                    plan.release(src, key)
                clock.lap("histograms_and_statistics")
        except Exception as e:  # noqa: BLE001 - every rank learns of it below
            err = e
        sharding.agree(err)
        stats = plan.gather(local, 7, np.float64)
        clock.lap("gather")
    if stats is not None:
        _result_rows(results, intervals, np.arange(len(intervals)), stats, lines)

    clock.lap("result_rows")
    if output_file is not None:
        if sharding.is_writer():
            writers.write_length_stats(output_file, results, short_reads, lines)
        else:
            writers.check_suffix(output_file, (".bed", ".bedgraph", ".bed.gz"), "The output file should have .bed or .bed.gz as as suffix.")
    clock.lap("write")
    clock.publish(LAST_STAGE_S)
    if verbose:
        stderr.write(f"Calculating fragment length statistics for intervals took {time.time() - t0} s\n")
    return results
