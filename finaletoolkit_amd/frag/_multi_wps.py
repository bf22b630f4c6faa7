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
from contextlib import nullcontext
from os import PathLike
from pathlib import Path
from sys import stderr, stdin
from typing import Union

import numpy as np

from .._stages import without_collector
from .. import sharding
from ..source import ContigFeed, get_engine, open_source
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


def _site_windows(site_bed, interval_size, lengths):
    """The windows ``multi_wps`` scores, from a site BED (behaviour of the reference's frag/_multi_wps.py:240-297):
    every site becomes ``interval_size`` bases around its midpoint, clipped to its contig (``lengths``: contig ->
    length; a site on another contig is skipped with a warning); a window that runs into the NEXT line's window on the
    same contig ends where that one starts; windows left empty are dropped.  Returns (contigs, starts, stops) in
    file order, as an object array and two int64 arrays."""
    reach = (round(-interval_size / 2), round(interval_size / 2))
    if reach[1] - reach[0] != interval_size:  # (round() sends both halves of an odd size to the even neighbour)
        raise AssertionError
    with (nullcontext(stdin) if site_bed == "-" else open(site_bed)) as bed:
        rows = [line.split() for line in bed]
    names, mids = [], []
    for fields in rows:  # line by line, so that warnings and the first error come in file order
        name, a, b = fields[0].strip(), int(fields[1]), int(fields[2])
        if a > b:
            raise ValueError(f"[multi_wps] {name}:{fields[1]}-{fields[2]} is invalid. Please be sure start coordinate "
                             f"occurs before stop for all intervals in {site_bed}.")
        if name not in lengths:
            warnings.warn(f"Skipping site {name}:{a} from site_bed (chrom not in chrom_sizes)", UserWarning)
            continue
        names.append(name)
        mids.append((a + b) // 2)
    names = np.array(names, dtype=object)
    mids = np.array(mids, dtype=np.int64)
    size = np.array([lengths[c] for c in names], dtype=np.int64)
    starts = np.maximum(mids + reach[0], 0)
    stops = np.minimum(mids + reach[1], size)
    if len(names) > 1:  # the next line's window, when it is on the same contig and starts inside this one, cuts it
        nxt = starts[1:]
        cut = (names[1:] == names[:-1]) & (nxt < stops[:-1])
        stops[:-1] = np.where(cut, nxt, stops[:-1])
    keep = stops > starts
    return names[keep], starts[keep], stops[keep]


@without_collector
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
    is_bam = isinstance(input_file, (str, PathLike)) and str(input_file).endswith((".sam", ".bam", ".cram"))
    one_process = sharding.rank_world()[1] == 1
    # several ranks, or a BAM (whose header is the contig list): the source is opened now; one process on a fragment
    # file: the decode starts below, once the sites say which contigs are wanted, and runs ahead of the scoring
    src = open_source(input_file, workers) if (is_bam or not one_process) else None
    eng = get_engine()
    header = _read_header(input_file, chrom_sizes, src)
    chrom_sizes_dict = dict(header)
    names, lo, hi = _site_windows(site_bed, interval_size, chrom_sizes_dict)
    if one_process:
        # (source.ContigFeed: a helper thread decodes towards the next contig while this one's windows are scored,
        # compressed and written; a lazily indexed file is read through its index for the sites' contigs alone)
        src = ContigFeed(input_file, workers, names=list(dict.fromkeys(str(c) for c in names)))
    # header order of the contigs, then start; equal keys keep their file order (the reference sorts the same way,
    # frag/_multi_wps.py:152-160)
    place = {chrom: k for k, (chrom, _) in enumerate(header)}
    order = np.lexsort((lo, np.array([place[c] for c in names], dtype=np.int64))) if len(names) else []
    contigs = [str(names[i]) for i in order]
    starts = [int(lo[i]) for i in order]
    stops = [int(hi[i]) for i in order]

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
    try:
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
        if isinstance(src, ContigFeed):
            src.finish()  # (a streamed file is decoded to its end, so that the cached source is whole)
    except BaseException:
        if isinstance(src, ContigFeed):
            src.close()
        raise
    if verbose:
        stderr.write(f"multi_wps took {time.time() - t0} s to complete\n")
    return output_file
