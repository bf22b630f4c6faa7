"""
Cleavage profile: per base, fragment ends over fragment depth in percent --
``cleavage_profile`` / ``multi_cleavage_profile`` with the reference's
signatures, interval merging and writers
(``src/finaletoolkit/frag/_cleavage_profile.py:93-500``).  Depth and ends
(``_coverage_and_ends``, ``:33-90``) come from the ``ftk_cleavage_intervals``
kernel, which shares the WPS tile skeleton.
"""
from __future__ import annotations

import time
import warnings
from pathlib import Path
from sys import stderr, stdin
from typing import Union

import numpy as np

from ..source import get_engine, open_source
from ..utils import chrom_sizes_to_dict, chrom_sizes_to_list
from ._runs import write_per_base_runs
from ._wps import _resolve_aliases

__all__ = ["cleavage_profile", "multi_cleavage_profile"]

_CLEAVAGE_DTYPE = [("contig", "U16"), ("pos", "i8"), ("proportion", "f8")]


def _result(contig, start, proportions):
    res = np.zeros(len(proportions), dtype=_CLEAVAGE_DTYPE)
    res["contig"] = contig
    res["pos"] = np.arange(start, start + len(proportions))
    res["proportion"] = proportions
    return res


def cleavage_profile(input_file, chrom_size: int, contig: str, start: int, stop: int, left: int = 0, right: int = 0,
                     min_length: int | None = None, max_length: int | None = None, quality_threshold: int = 30,
                     verbose: Union[bool, int] = 0, fraction_low: int | None = None,
                     fraction_high: int | None = None, reference_file: str | Path | None = None) -> np.ndarray:
    """Structured array ``('contig', 'pos', 'proportion')`` over ``[start-left, stop+right)`` clipped to the contig."""
    if verbose:
        t0 = time.time()
    min_length, max_length = _resolve_aliases(min_length, max_length, fraction_low, fraction_high)
    adj_start = max(start - left, 0)
    adj_stop = min(stop + right, chrom_size)
    src = open_source(input_file)
    props = get_engine().cleavage(src.require_interval(contig, adj_start, adj_stop, 1), adj_start, adj_stop, min_length, max_length, quality_threshold)
    if verbose:
        stderr.write(f"cleavage_profile took {time.time() - t0} s to complete\n")
    return _result(contig, adj_start, props)


def _read_intervals(interval_file, left, right, chrom_dict):
    """Expand by left/right, clip, merge overlapping neighbours (frag/_cleavage_profile.py:411-451)."""
    merged: list[list] = []
    bed = stdin if interval_file == "-" else open(interval_file)
    try:
        for line in bed:
            fields = line.split()
            contig = fields[0].strip()
            start, stop = int(fields[1]), int(fields[2])
            if contig not in chrom_dict:
                warnings.warn(f"Skipping interval {contig}:{start}-{stop} from interval_file "
                              f"({contig} not in chrom_sizes)", UserWarning)
                continue
            start = max(0, start - left)
            stop = min(stop + right, chrom_dict[contig])
            if merged and merged[-1][0] == contig and start < merged[-1][2]:
                merged[-1][2] = max(merged[-1][2], stop)
            else:
                merged.append([contig, start, stop])
    finally:
        if interval_file != "-":
            bed.close()
    return [m[0] for m in merged], [m[1] for m in merged], [m[2] for m in merged]


def multi_cleavage_profile(input_file, interval_file, chrom_sizes, left: int = 0, right: int = 0,
                           min_length: int | None = None, max_length: int | None = None,
                           quality_threshold: int = 30, output_file: str = "-", workers: int = 1,
                           verbose: Union[bool, int] = 0, fraction_low: int | None = None,
                           fraction_high: int | None = None, reference_file: str | Path | None = None) -> str:
    """Cleavage profile over the (padded, merged) intervals of a BED file -> ``.bw`` / ``.bed.gz`` / ``bedgraph.gz``."""
    if verbose:
        t0 = time.time()
    min_length, max_length = _resolve_aliases(min_length, max_length, fraction_low, fraction_high)
    if input_file == "-" and interval_file == "-":
        raise ValueError("input_file and site_bed cannot both read from stdin")
    if chrom_sizes is None:
        raise ValueError("chrom_sizes must be specified.")
    header = chrom_sizes_to_list(chrom_sizes)
    contigs, starts, stops = _read_intervals(interval_file, left, right, chrom_sizes_to_dict(chrom_sizes))
    src = open_source(input_file, workers)
    eng = get_engine()

    def score_run(key, c, run_starts, run_stops):  # one launch per unit of intervals on the same contig
        return eng.cleavage_intervals(key, run_starts, run_stops, min_length, max_length,
                                      quality_threshold)

    # Pool(workers) of the reference (:372-395) = one rank per GPU here: the intervals cut into equal-cost shares over
    # the ranks (frag/_runs.py; a partial share of a contig scored from a region of it), rank 0 writes
    if isinstance(output_file, str):
        if output_file.endswith(".bw"):
            write_per_base_runs(output_file, "bw", header, contigs, starts, stops, score_run, src, 1)
        elif output_file.endswith(".bed.gz") or output_file.endswith("bedgraph.gz") or output_file == "-":
            # rows "contig  pos  pos+1  proportion" with the floats printed as Python prints them, formatted by
            # the library's host threads; gzip members compressed in parallel
            write_per_base_runs(output_file, "bedgraph.gz", header, contigs, starts, stops, score_run, src, 1)
        else:
            raise ValueError("output_file can only have suffix .bw, .bedgraph.gz, or .bed.gz.")
    elif output_file is not None:
        raise TypeError(f'output_file is unsupported type "{type(input_file)}". output_file should be a string '
                        "specifying the path of the file to output scores to.")
    if verbose:
        stderr.write(f"cleavage profile took {time.time() - t0} s to complete\n")
    return output_file
