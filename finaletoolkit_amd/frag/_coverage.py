"""
Fragment coverage over intervals -- ``single_coverage`` / ``coverage`` /
``CoverageResult`` with the reference's signatures, defaults, return types,
output formats and errors (``src/finaletoolkit/frag/_coverage.py:26-305``).

The per-window count loop (``:117-130``) runs as one ``ftk_window_counts``
launch per contig over every interval of the BED file at once.
"""
from __future__ import annotations

import sys
import time
from pathlib import Path
from typing import NamedTuple, Union

import numpy as np

from .. import sharding, writers
from .._stages import Stages, without_collector
from ..source import ContigFeed, get_engine, open_source
from ..utils import _check_policy, _check_region, _region_contigs, get_intervals

__all__ = ["coverage", "single_coverage", "CoverageResult"]


LAST_STAGE_S: dict = {}  # the last coverage() call's wall time by stage (seconds)


class CoverageResult(NamedTuple):
    """``(contig, start, stop, name, coverage)`` (frag/_coverage.py:26-50)."""

    contig: str | None
    start: int | None
    stop: int | None
    name: str
    coverage: float


def _owners(src, names, extent=None):
    """Rank of every contig of ``names`` under the initialised process group (LPT on a cost estimate: the
    contig's length when the file knows it -- BAM header -- else the furthest interval stop, else 1) and
    the weights used.  One map per call, shared by the genome-wide total and the interval counts, so a rank
    decodes only the contigs it owns."""
    rank, world = sharding.rank_world()
    weights = {c: float(src.lengths.get(c) or (extent or {}).get(c) or 1) for c in names}
    return rank, world, sharding.lpt_assign(weights, world), weights


def _total(src, contig, start, stop, min_length, max_length, intersect_policy, quality_threshold,
           shard=None) -> int:
    """Fragments of the region (the whole file when ``contig`` is None).  ``shard = (rank, world, owner)``:
    every rank sums the contigs it owns and the sums meet in one int64 all-reduce."""
    eng = get_engine()
    if shard is not None and contig is None and shard[1] > 1:
        rank, world, owner = shard
        total = 0
        for c in src.contigs:  # no load_all: a rank decodes only what it owns
            if owner.get(c, 0) == rank and src.has(c):
                total += int(eng.window_counts(src.require(c), [None], [None], quality_threshold, min_length,
                                               max_length, intersect_policy)[0])
        return sharding.allreduce_sum(total)
    names, whole = _region_contigs(src, contig)
    total = 0
    for c in names:
        total += int(eng.window_counts(src.require(c) if whole else src.require_interval(c, start, stop, 1), [None if whole else start], [None if whole else stop],
                                       quality_threshold, min_length, max_length, intersect_policy)[0])
    return total


def single_coverage(input_file: Union[str, Path], contig: str | None = None, start: int | None = 0,
                    stop: int | None = None, name: str | None = ".", min_length: int | None = None,
                    max_length: int | None = None, intersect_policy: str = "midpoint",
                    quality_threshold: int = 30, verbose: bool | int = False,
                    reference_file: str | Path | None = None) -> CoverageResult:
    """Count fragments (per ``intersect_policy``) in ``contig:[start, stop)``."""
    if verbose:
        t0 = time.time()
        sys.stderr.write(f"single_coverage: {input_file} {contig}:{start}-{stop}\n")
    _check_policy(intersect_policy)
    _check_region(contig, start, stop)
    src = open_source(input_file)
    src.check_fetch(contig, start, stop)
    cov = _total(src, contig, start, stop, min_length, max_length, intersect_policy, quality_threshold)
    if verbose:
        sys.stderr.write(f"single_coverage took {time.time() - t0} s to complete\n")
    return CoverageResult(contig, start, stop, "." if name is None else name, cov)


def _by_contig(intervals):
    by_contig: dict[str, list[int]] = {}
    extent: dict[str, int] = {}
    for i, (c, _, b, _) in enumerate(intervals):
        by_contig.setdefault(c, []).append(i)
        extent[c] = max(extent.get(c, 0), int(b))
    return by_contig, extent


def _interval_counts(src, intervals, min_length, max_length, intersect_policy, quality_threshold):
    """Counts for every interval, in interval order (the ``imap`` of :244-248).  The reference spreads the
    intervals over ``Pool(workers)``; here the intervals are cut into equal-cost runs over the ranks of the process
    group (``sharding.IntervalPlan``: whole contigs, a region of the contig a cut falls into), every rank counts its
    share with one launch per unit and one all-gather hands every rank the full vector."""
    eng = get_engine()
    plan = sharding.IntervalPlan([iv[0] for iv in intervals], [iv[1] for iv in intervals], [iv[2] for iv in intervals])
    local, err = {}, None
    try:
        for unit in plan.mine:
            idx = plan.intervals(unit)
            key = plan.unit_key(src, unit, 1)
            try:
                local[unit] = eng.window_counts(key, plan.starts[idx].astype(np.int32), plan.stops[idx].astype(np.int32),
                                                quality_threshold, min_length, max_length, intersect_policy).reshape(-1, 1)
            finally:
                plan.release(src, key)
    except Exception as e:  # noqa: BLE001 - with several ranks every rank must learn of it
        err = e
    sharding.agree(err)
    return plan.gather(local, 1).reshape(-1)


def _coverage_one_process(clock, input_file, interval_file, min_length, max_length, normalize, intersect_policy,
                          quality_threshold, workers):
    """One process (no rank group): the decode starts at once on a helper thread (``source.ContigFeed``), the interval
    file is read beside it, and every contig is counted - its intervals in one launch, its total for ``normalize`` in
    another - as soon as it is resident, while the decoder is in the next one.  Returns ``(counts, total, intervals)``
    with ``counts`` in interval order (the ``imap`` of the reference, frag/_coverage.py:244-248)."""
    eng = get_engine()
    # ``normalize`` counts the whole file, so every contig is wanted and the decode can start before the intervals are
    # known; otherwise only the intervals' contigs are (a lazily indexed file is then read through its index for them)
    feed = ContigFeed(input_file, workers) if normalize else None
    intervals = get_intervals(interval_file)
    by_contig, _ = _by_contig(intervals)
    clock.lap("read_intervals")
    if feed is None:
        feed = ContigFeed(input_file, workers, names=list(by_contig))
    starts = np.array([iv[1] for iv in intervals], dtype=np.int64)
    stops = np.array([iv[2] for iv in intervals], dtype=np.int64)
    counts = np.zeros(len(intervals), np.int64)
    total, left = 0, dict(by_contig)
    try:
        for src, c in feed:
            clock.lap("decode_wait")
            key = src.key(c)
            if normalize:  # single_coverage(input_file, None, 0, None): every fragment of the file once (:215-227)
                total += int(eng.window_counts(key, [None], [None], quality_threshold, min_length, max_length,
                                               intersect_policy)[0])
            idx = left.pop(c, None)
            if idx is not None:
                idx = np.asarray(idx, dtype=np.int64)
                order = idx[np.argsort(starts[idx], kind="stable")]  # (sorted windows take the block-per-window launch)
                counts[order] = eng.window_counts(key, starts[order].astype(np.int32), stops[order].astype(np.int32),
                                                  quality_threshold, min_length, max_length, intersect_policy)
            clock.lap("count_kernels")
        src = feed.finish()
    except BaseException:
        feed.close()
        raise
    for c in left:  # contigs the file does not hold: the error the reference's fetch raises (ValueError)
        src.require(c)
    return counts, total, intervals


@without_collector
def coverage(input_file: Union[str, Path], interval_file: str, output_file: str, scale_factor: float = 1.0,
             min_length: int | None = None, max_length: int | None = None, normalize: bool = False,
             intersect_policy: str = "midpoint", quality_threshold: int = 30, workers: int = 1,
             verbose: Union[bool, int] = False, reference_file: str | Path | None = None) -> list[CoverageResult]:
    """Coverage of every BED interval; optional genome-wide normalisation
    (frag/_coverage.py:145-305).  ``workers`` sizes the decoder thread pool."""
    if verbose:
        t0 = time.time()
        sys.stderr.write(f"coverage: {input_file} over {interval_file}\n")
    _check_policy(intersect_policy)
    clock = Stages()
    if sharding.rank_world()[1] == 1:
        counts, total, intervals = _coverage_one_process(clock, input_file, interval_file, min_length, max_length, normalize,
                                                         intersect_policy, quality_threshold, workers)
    else:
        src = open_source(input_file, workers)
        clock.lap("open")
        intervals = get_intervals(interval_file)
        clock.lap("read_intervals")
        if normalize:
            # single_coverage(input_file, None, 0, None, "."): the whole file (:215-227).  Every fragment counts once, so
            # this part is dealt by whole contigs (LPT on their lengths; a region read returns a fragment to every rank
            # whose region it overlaps), one int64 all-reduce.
            by_contig, extent = _by_contig(intervals)
            names = list(dict.fromkeys(list(src.contigs) + list(by_contig)))
            rank, world, owner, _weights = _owners(src, names, extent)
            total = _total(src, None, 0, None, min_length, max_length, intersect_policy, quality_threshold,
                           shard=(rank, world, owner))
            clock.lap("decode_and_total")
        counts = _interval_counts(src, intervals, min_length, max_length, intersect_policy, quality_threshold)
        clock.lap("interval_counts")
    if normalize:
        if verbose:
            sys.stderr.write(f"Total coverage is {total}\n")
        scale_factor /= total

    values = (counts * scale_factor).tolist()  # int64 -> float64 times a Python float: `cov * scale_factor` of :250
    import gc
    was = gc.isenabled()
    gc.disable()  # (tens of thousands of small tuples: the collector would walk them again and again)
    try:
        return_val = [CoverageResult(c, a, b, n, v) for (c, a, b, n), v in zip(intervals, values)]
    finally:
        if was:
            gc.enable()
    if output_file is not None:
        if sharding.rank_world()[0] == 0:
            writers.write_coverage_rows(output_file, intervals, values)
        else:  # every rank returns the same list; rank 0 alone writes the file / stdout
            writers.check_suffix(output_file, (".bed", ".bedgraph", ".bed.gz"), "output_file should have .bed or .bed.gz as suffix")
    clock.lap("rows_and_write")
    clock.publish(LAST_STAGE_S)
    if verbose:
        sys.stderr.write(f"coverage took {time.time() - t0} s to complete\n")
    return return_val
