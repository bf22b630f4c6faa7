"""
WPS over many BED sites -- ``multi_wps`` with the reference's signature and
site handling (``src/finaletoolkit/frag/_multi_wps.py:31-341``): each site is
replaced by an ``interval_size`` window centred on its midpoint, clipped to the
contig, the previous window truncated where the next one starts, intervals
sorted into header order.  All windows of a contig are scored in ONE
``ftk_wps_intervals`` launch instead of a process pool of ``wps`` calls; with
several ranks (one per GPU) the contigs are dealt to the ranks and rank 0 writes.
"""
from __future__ import annotations

import time
import warnings
from os import PathLike
from pathlib import Path
from sys import stderr, stdin
from typing import Union

import numpy as np

from ..source import get_engine, open_source
from ..utils import chrom_sizes_to_list
from ._runs import write_per_base_runs
from ._wps import _resolve_aliases

__all__ = ["multi_wps"]


def _read_header(input_file, chrom_sizes, src) -> list[tuple[str, int]]:
    """(contig, length) pairs from the BAM header or chrom.sizes (frag/_multi_wps.py:226-237)."""
    if isinstance(input_file, (str, PathLike)) and str(input_file).endswith((".sam", ".bam", ".cram")):
        return [(c, int(src.lengths[c])) for c in src.contigs]
    if chrom_sizes is None:
        raise ValueError("chrom_sizes must be specified for BED/Fragment files")
    return chrom_sizes_to_list(chrom_sizes)


def _read_sites(site_bed, interval_size, references, chrom_sizes_dict):
    """Centred, non-overlapping windows from a site BED (frag/_multi_wps.py:240-297)."""
    contigs, starts, stops = [], [], []
    left_of_site = round(-interval_size / 2)
    right_of_site = round(interval_size / 2)
    assert right_of_site - left_of_site == interval_size
    bed = stdin if site_bed == "-" else open(site_bed)
    try:
        prev_contig = None
        prev_start = 0
        prev_stop = 0
        for line in bed:
            contents = line.split()
            contig = contents[0].strip()
            if int(contents[1]) > int(contents[2]):
                raise ValueError(
                    f"[multi_wps] {contig}:{contents[1]}-{contents[2]} is invalid. Please be sure start coordinate "
                    f"occurs before stop for all intervals in {site_bed}.")
            if contig not in references:
                warnings.warn(f"Skipping site {contig}:{int(contents[1])} from site_bed (chrom not in chrom_sizes)",
                              UserWarning)
                continue
            midpoint = (int(contents[1]) + int(contents[2])) // 2
            start = max(0, midpoint + int(left_of_site))
            stop = min(midpoint + int(right_of_site), chrom_sizes_dict[contig])
            if contig == prev_contig and start < prev_stop:
                prev_stop = start
            if prev_contig is not None and prev_stop > prev_start:
                contigs.append(prev_contig)
                starts.append(prev_start)
                stops.append(prev_stop)
            prev_contig = contig
            prev_start = start
            prev_stop = stop
        if prev_stop > prev_start:
            contigs.append(prev_contig)
            starts.append(prev_start)
            stops.append(prev_stop)
    finally:
        if site_bed != "-":
            bed.close()
    return contigs, starts, stops


def multi_wps(input_file, site_bed, chrom_sizes=None, output_file: str | None = None, window_size: int = 120,
              interval_size: int = 5000, min_length: int = 120, max_length: int = 180, quality_threshold: int = 30,
              workers: int = 1, verbose: Union[bool, int] = 0, fraction_low: int | None = None,
              fraction_high: int | None = None, reference_file: str | Path | None = None) -> str | None:
    """Aggregate WPS over the sites of a BED file; writes ``.bw`` or
    ``.bed.gz``/``bedGraph.gz`` and returns the output path."""
    if verbose:
        t0 = time.time()
        stderr.write(f"Calculating aggregate WPS: {input_file} {site_bed} -> {output_file}\n")
    if input_file == "-" and site_bed == "-":
        raise ValueError("input_file and site_bed cannot both read from stdin")
    min_length, max_length = _resolve_aliases(min_length, max_length, fraction_low, fraction_high)
    src = open_source(input_file, workers)
    eng = get_engine()
    header = _read_header(input_file, chrom_sizes, src)
    references = [chrom for (chrom, _) in header]
    chrom_sizes_dict = dict(header)
    contigs, starts, stops = _read_sites(site_bed, interval_size, references, chrom_sizes_dict)

    if header and contigs:  # header (contig) order, then start (:152-160)
        chrom_order = {chrom: idx for idx, (chrom, _) in enumerate(header)}
        order = sorted(range(len(contigs)), key=lambda i: (chrom_order.get(contigs[i], len(header)), starts[i]))
        contigs = [contigs[i] for i in order]
        starts = [starts[i] for i in order]
        stops = [stops[i] for i in order]
    try:
        [chrom_sizes_dict[c] for c in contigs]
    except KeyError as e:
        raise ValueError(f"Chrom {e} from {site_bed} is not present in {input_file} or chrom.sizes file if "
                         "applicable). Please ensure that all files use the same reference genome and chromosome "
                         "naming conventions.")

    def score_run(key, c, run_starts, run_stops):
        """All intervals of one unit in ONE launch; interval k of the unit is values[offsets[k]:offsets[k+1]]."""
        return eng.wps_intervals(key, run_starts, run_stops, chrom_sizes_dict[c], int(window_size),
                                 0 if min_length is None else int(min_length), int(max_length),
                                 int(quality_threshold))

    # The reference scores the intervals in Pool(workers) and the parent writes them in order (:196-198,
    # :300-341).  Here the intervals are cut into equal-cost consecutive shares over the ranks of the process group (one
    # per GPU; a partial share of a contig is scored from a region of it), every rank scores, formats and compresses
    # its own, rank 0 lays the pieces into the file (frag/_runs.py).
    # rows a share can need: the reference's fetch window of an interval is [start - max_length, stop + max_length)
    # (frag/_wps.py:156-157), and for a BAM the read1 ALIGNMENTS overlapping it decide
    pad = max(int(window_size), int(max_length)) + 1
    if isinstance(output_file, str):
        if output_file.endswith(".bw"):
            write_per_base_runs(output_file, "bw", header, contigs, starts, stops, score_run, src, pad)
        elif output_file.endswith(".bed.gz") or output_file.endswith("bedGraph.gz"):
            write_per_base_runs(output_file, "bedgraph.gz", header, contigs, starts, stops, score_run, src, pad)
        else:
            raise ValueError("output_file can only have suffix .bw")
    elif output_file is not None:
        raise TypeError(f'output_file is unsupported type "{type(input_file)}". output_file should be a string '
                        "specifying the path of the file to output scores to.")
    if verbose:
        stderr.write(f"multi_wps took {time.time() - t0} s to complete\n")
    return output_file
