"""
Breakpoint-motif features with the reference's surface
(``src/finaletoolkit/frag/_breakpoint_motifs.py``): the k-mer is read symmetrically around each
breakpoint, ``ref[start-k//2 : start+k//2]`` and the reverse complement of
``ref[stop-k//2 : stop+k//2]``; fragments whose start lies within ``k//2`` of either contig end
are dropped (:127-135).  Counting runs on the GPU (``ftk_motif_counts``).

An odd ``k`` reads ``2*(k//2) != k`` bases, which the reference rejects fragment by fragment
(:146-153): every count stays zero.  The per-fragment warnings of the reference are not emitted.
"""
from __future__ import annotations

import time
from sys import stderr, stdout

import numpy as np

from ..reference import ReferenceGenome
from ._motif_common import (MotifFreqs, MotifsIntervals, gen_kmers, genome_windows, parse_intervals_arg,
                            region_histograms, resolve_motif_aliases, write_motif_freqs)

__all__ = ["BreakpointMotifFreqs", "BreakpointMotifsIntervals", "region_breakpoint_motifs", "breakpoint_motifs",
           "interval_breakpoint_motifs"]


class BreakpointMotifFreqs(MotifFreqs):
    """Genome-wide breakpoint-motif k-mer frequencies."""

    def __init__(self, kmer_frequencies, k, quality_threshold: int = 30):
        super().__init__(kmer_frequencies, k, quality_threshold)


class BreakpointMotifsIntervals(MotifsIntervals):
    """Interval-stratified breakpoint-motif k-mer counts."""

    def __init__(self, intervals, k, quality_threshold: int = 30):
        super().__init__(intervals, k, quality_threshold)


def _histograms(input_file, refseq_file, regions, k, both_strands, negative_strand, quality_threshold, workers=1):
    if both_strands and negative_strand:
        raise ValueError("Cannot have both both_strands and negative_strand.")
    if k % 2 or k < 2:
        return np.zeros((len(regions), 4 ** k), np.uint32)
    h = k // 2
    spec = dict(k=k, fwd_offset=-h, rev_offset=-h, both_strands=both_strands, negative_strand=negative_strand,
                guard=h, rev_oob_is_error=False)
    return region_histograms(input_file, refseq_file, regions, spec, quality_threshold, workers)


def region_breakpoint_motifs(input_file, contig, start, stop, refseq_file, k: int = 6, fraction_low: int = 10,
                             fraction_high: int = 600, both_strands: bool = True, negative_strand: bool = False,
                             output_file=None, quality_threshold: int = 30, verbose=False) -> dict:
    """k-mer -> count for the fragments fetched for ``contig:start-stop`` (all ``4**k`` keys present)."""
    t0 = time.time()
    counts = _histograms(input_file, refseq_file, [(contig, start, stop)], k, both_strands, negative_strand,
                         quality_threshold)[0]
    if verbose:
        stderr.write(f"region_breakpoint_motifs took {time.time() - t0} seconds to run\n")
    return dict(zip(gen_kmers(k), counts.tolist()))


def breakpoint_motifs(input_file, refseq_file, k: int = 6, min_length: int = 50, max_length: int = None,
                      both_strands: bool = True, negative_strand: bool = False, output_file=None,
                      quality_threshold: int = 30, workers: int = 1, verbose=False, fraction_low=None,
                      fraction_high=None) -> BreakpointMotifFreqs:
    """Genome-wide breakpoint-motif frequencies, summed over the 1 Mb windows of every reference contig."""
    t0 = time.time()
    resolve_motif_aliases(min_length, max_length, fraction_low, fraction_high)
    with ReferenceGenome(refseq_file) as ref:
        windows = genome_windows(ref.chroms)
    hist = _histograms(input_file, refseq_file, windows, k, both_strands, negative_strand, quality_threshold, workers)
    total = hist.sum(axis=0, dtype=np.float64)
    results = BreakpointMotifFreqs(zip(gen_kmers(k), total / np.sum(total)), k, quality_threshold)
    write_motif_freqs(results, output_file)
    if verbose:
        stdout.write(f"breakpoint_motifs took {time.time() - t0} seconds to run\n")
    return results


def interval_breakpoint_motifs(input_file, refseq_file, intervals, k: int = 6, min_length=50, max_length=None,
                               both_strands: bool = True, negative_strand: bool = False, output_file=None,
                               quality_threshold: int = 30, workers: int = 1, verbose=False, fraction_low=None,
                               fraction_high=None) -> BreakpointMotifsIntervals:
    """Breakpoint-motif counts for each interval of a BED file / list of ``(chrom, start, stop, name)``."""
    t0 = time.time()
    resolve_motif_aliases(min_length, max_length, fraction_low, fraction_high)
    tuples = parse_intervals_arg(intervals)
    hist = _histograms(input_file, refseq_file, tuples, k, both_strands, negative_strand, quality_threshold, workers)
    kmers = gen_kmers(k)
    results = BreakpointMotifsIntervals([(iv, dict(zip(kmers, row.tolist()))) for iv, row in zip(tuples, hist)], k,
                                        quality_threshold)
    write_motif_freqs(results, output_file)
    if verbose:
        stdout.write(f"breakpoint_motifs took {time.time() - t0} seconds to run\n")
    return results
