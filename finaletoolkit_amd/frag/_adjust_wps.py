"""
``adjust_wps`` with the reference's signature (``src/finaletoolkit/frag/_adjust_wps.py:161-330``):
raw WPS from a bigWig, per interval minus its running median (or mean), then a
Savitzky-Golay pass, written back as a bigWig.

The reference fans the intervals out to a process pool, each worker building a
``(len - W) x W`` sliding-window matrix for ``np.median``.  Here the intervals are
read once, laid end to end, and filtered on the GPU in batches by
``ftk_wps_adjust`` (exact rank-selection median in LDS, then the Savitzky-Golay
taps); interval parsing, merging rules, the skip/raise behaviour and the output
layout follow the reference.
"""
from __future__ import annotations

import gzip
import time
from sys import stderr
from typing import Union

import numpy as np

from ..bigwig import BigWigFile, write_fixed_step_bigwig
from ..source import get_engine
from ..utils import chrom_sizes_to_list

__all__ = ["adjust_wps"]

# scores handed to one ftk_wps_adjust call (float64 each): bounds host staging to ~0.5 GB
_BATCH_SCORES = 1 << 25


def _read_intervals(interval_file, interval_size, median_window_size):
    """Centred intervals, merged where the median filter's trimmed ends would still overlap
    (frag/_adjust_wps.py:226-262)."""
    # reach of an interval either side of its site's midpoint; Python's round() sends both halves of an odd size to the
    # even neighbour, so an odd interval_size cannot be met - the reference asserts that (frag/_adjust_wps.py:219-221)
    reach = (round(-interval_size / 2), round(interval_size / 2))
    if reach[1] - reach[0] != interval_size:
        raise AssertionError
    if not (interval_file.endswith(".bed") or interval_file.endswith(".bed.gz")):
        raise ValueError("Invalid filetype for interval_file.")
    end_decrease = median_window_size // 2
    out: list[tuple[str, int, int]] = []
    opener = gzip.open if interval_file.endswith(".gz") else open
    with opener(interval_file, "rt") as fh:
        for line in fh:
            f = line.split("\t")
            contig = f[0].strip()
            mid = (int(f[1]) + int(f[2])) // 2
            start = max(0, mid + reach[0])
            stop = mid + reach[1]
            if out and out[-1][0] == contig and out[-1][2] - end_decrease > start + end_decrease:
                start = out[-1][1]
                out.pop()
            out.append((contig, int(start), int(stop)))
    return out


def _load_scores(bw, contig, start, stop):
    """Positions and scores of one interval, or None to skip it
    (frag/_adjust_wps.py:79-117 and the RuntimeError branch :146-154)."""
    try:
        got = bw.intervals(contig, start, stop)
    except RuntimeError as e:
        stderr.write(f"{type(e).__name__}: {e}\nInvalid interval detected:\n{contig}:{start}-{stop}. "
                     "This interval will be skipped.\n")
        return None
    if got is None:
        stderr.write(f"No entries in range: {contig}:{start}-{stop}. This interval will be skipped.\n")
        return None
    starts, _, scores = got
    if len(starts) > 1 and not np.all(starts[:-1] + 1 == starts[1:]):
        raise ValueError(
            "BigWig was found to be nonsequential. There may be multiple entries for one position or gaps in the "
            "regions specified in the interval file.")
    return starts, scores


def adjust_wps(
    input_file: str,
    interval_file: str,
    output_file: str,
    chrom_sizes: str,
    interval_size: int = 5000,
    median_window_size: int = 1000,
    savgol_window_size: int = 21,
    savgol_poly_deg: int = 2,
    savgol: bool = True,
    mean: bool = False,
    subtract_edges: bool = False,
    edge_size: int = 500,
    workers: int = 1,
    verbose: Union[bool, int] = False,
) -> None:
    """Adjust raw WPS in a bigWig with median/mean and Savitzky-Golay filters
    (arguments as in the reference; ``workers`` is accepted and unused: the filter runs on the GPU)."""
    t0 = time.time()
    if verbose:
        stderr.write("Reading intervals from bed...\n")
    intervals = _read_intervals(interval_file, interval_size, median_window_size)
    if not str(input_file).endswith(".bw"):
        raise ValueError("Invalid filetype for input_file.")
    header = chrom_sizes_to_list(chrom_sizes)
    eng = get_engine()
    W = int(median_window_size)
    results = []  # (contig, first adjusted position, values)

    def flush(batch):
        if not batch:
            return
        offs = np.zeros(len(batch) + 1, np.int64)
        np.cumsum([len(b[2]) for b in batch], out=offs[1:])
        scores = np.concatenate([b[2] for b in batch])
        sub = np.array([b[3] for b in batch], np.float64) if subtract_edges else None
        out = eng.wps_adjust(scores, offs, W, mean, sub, savgol_window_size, savgol_poly_deg, savgol)
        for i, (contig, starts, _, _) in enumerate(batch):
            lo = int(offs[i]) - i * W
            n = int(offs[i + 1] - offs[i]) - W
            results.append((contig, int(starts[W // 2]) if n > 0 else 0, out[lo:lo + n]))

    with BigWigFile(input_file) as bw:
        batch, held = [], 0
        for contig, start, stop in intervals:
            got = _load_scores(bw, contig, start, stop)
            if got is None:
                continue
            starts, scores = got
            sub = 0.0
            if subtract_edges:
                sub = np.mean([np.mean(scores[:edge_size]), np.mean(scores[-edge_size:])])
            if W > len(scores):
                raise ValueError(f"median_window_size ({W}) cannot be greater than the length of interval "
                                 f"({len(scores)}).")
            batch.append((contig, starts, scores, sub))
            held += len(scores)
            if held >= _BATCH_SCORES:
                flush(batch)
                batch, held = [], 0
        flush(batch)

    if verbose:
        stderr.write("Writing to output\n")
    write_fixed_step_bigwig(output_file, header, ((c, s, v) for c, s, v in results if len(v)))
    if verbose:
        stderr.write(f"Adjust-WPS took {time.time() - t0} s to run.\n")
