"""
Fragment coverage over intervals -- ``single_coverage`` / ``coverage`` /
``CoverageResult`` with the reference's signatures, defaults, return types,
output formats and errors (``src/finaletoolkit/frag/_coverage.py:26-305``).

The per-window count loop (``:117-130``) runs as one ``ftk_window_counts``
launch per contig over every interval of the BED file at once.
"""
from __future__ import annotations

import gzip
import sys
import time
from pathlib import Path
from typing import NamedTuple, Union

import numpy as np

from .. import sharding
from ..source import get_engine, open_source
from ..utils import _check_policy, _check_region, _region_contigs, get_intervals

__all__ = ["coverage", "single_coverage", "CoverageResult"]


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
    src = open_source(input_file, workers)
    intervals = get_intervals(interval_file)
    if normalize:
        # single_coverage(input_file, None, 0, None, "."): the whole file (:215-227).  Every fragment counts once, so
        # this part is dealt by whole contigs (LPT on their lengths; a region read returns a fragment to every rank
        # whose region it overlaps), one int64 all-reduce.
        by_contig, extent = _by_contig(intervals)
        names = list(dict.fromkeys(list(src.contigs) + list(by_contig)))
        rank, world, owner, _weights = _owners(src, names, extent)
        total = _total(src, None, 0, None, min_length, max_length, intersect_policy, quality_threshold,
                       shard=(rank, world, owner))
    counts = _interval_counts(src, intervals, min_length, max_length, intersect_policy, quality_threshold)
    if normalize:
        if verbose:
            sys.stderr.write(f"Total coverage is {total}\n")
        scale_factor /= total

    if output_file is not None and sharding.rank_world()[0] != 0:
        # every rank returns the same list; rank 0 alone writes the file / stdout
        if not (output_file.endswith((".bed", ".bedgraph", ".bed.gz")) or output_file == "-"):
            raise ValueError("output_file should have .bed or .bed.gz as suffix")
        output_file = None
    return_val: list[CoverageResult] = []
    output_is_file = False
    if output_file is not None:
        try:
            if output_file.endswith(".bed") or output_file.endswith(".bedgraph"):
                output_is_file = True
                output = open(output_file, "w")
            elif output_file.endswith(".bed.gz"):
                output = gzip.open(output_file, "wt")
                output_is_file = True
            elif output_file == "-":
                output = sys.stdout
            else:
                raise ValueError("output_file should have .bed or .bed.gz as suffix")
            bedgraph = output_file.endswith(".bedgraph")
            for (contig, start, stop, name), cov in zip(intervals, counts.tolist()):
                value = cov * scale_factor
                if bedgraph:
                    output.write(f"{contig}\t{start}\t{stop}\t{value}\n")
                else:
                    output.write(f"{contig}\t{start}\t{stop}\t{name}\t{value}\n")
                return_val.append(CoverageResult(contig, start, stop, name, value))
        finally:
            if output_is_file:
                output.close()
    else:
        return_val = [CoverageResult(contig, start, stop, name, cov * scale_factor)
                      for (contig, start, stop, name), cov in zip(intervals, counts.tolist())]
    if verbose:
        sys.stderr.write(f"coverage took {time.time() - t0} s to complete\n")
    return return_val
