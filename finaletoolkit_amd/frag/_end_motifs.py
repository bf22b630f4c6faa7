"""
5' end-motif features with the reference's surface (``src/finaletoolkit/frag/_end_motifs.py``):
``region_end_motifs``, ``end_motifs``, ``interval_end_motifs``, ``EndMotifFreqs``,
``EndMotifsIntervals``.  The per-fragment reference lookups run on the GPU
(``ftk_motif_counts``): the forward k-mer is ``ref[start:start+k]``, the reverse one the reverse
complement of ``ref[stop-k:stop]``.

As in the reference, the fragment-length arguments only trigger the ``< k`` warning: the region
worker fetches by overlap and mapping quality alone (frag/_end_motifs.py:112-118).
"""
from __future__ import annotations

import time
import warnings
from sys import stderr, stdout

import numpy as np

from ..reference import ReferenceGenome
from ._motif_common import (MIN_QUALITY, MotifFreqs, MotifsIntervals, gen_kmers, genome_windows, parse_intervals_arg,
                            region_histograms, resolve_motif_aliases, write_motif_freqs)

__all__ = ["EndMotifFreqs", "EndMotifsIntervals", "region_end_motifs", "end_motifs", "interval_end_motifs",
           "MIN_QUALITY"]


class EndMotifFreqs(MotifFreqs):
    """Genome-wide 5' end-motif k-mer frequencies (Zhou et al., 2023)."""


class EndMotifsIntervals(MotifsIntervals):
    """Interval-stratified 5' end-motif k-mer counts."""


def _spec(k, both_strands, negative_strand):
    if both_strands and negative_strand:
        raise ValueError("Cannot have both both_strands and negative_strand.")
    return dict(k=k, fwd_offset=0, rev_offset=-k, both_strands=both_strands, negative_strand=negative_strand,
                guard=0, rev_oob_is_error=both_strands)


def _clamp_min(value, k, name):
    if value is not None and value < k:
        warnings.warn(f"{name}={value} < k={k}, which may cause errors. Automatically setting {name}=k.")
        return k
    return value


def region_end_motifs(input_file, contig, start, stop, refseq_file, k: int = 4, fraction_low=50,
                      fraction_high=None, both_strands: bool = True, negative_strand: bool = False,
                      output_file=None, quality_threshold: int = MIN_QUALITY, verbose=False) -> dict:
    """k-mer -> count for the fragments fetched for ``contig:start-stop`` (all ``4**k`` keys present)."""
    t0 = time.time()
    spec = _spec(k, both_strands, negative_strand)
    _clamp_min(fraction_low, k, "fraction_low")
    counts = region_histograms(input_file, refseq_file, [(contig, start, stop)], spec, quality_threshold)[0]
    if verbose:
        stderr.write(f"region_end_motifs took {time.time() - t0} seconds to run\n")
    return dict(zip(gen_kmers(k), counts.tolist()))


def end_motifs(input_file, refseq_file, k: int = 4, min_length=50, max_length=None, both_strands: bool = True,
               negative_strand: bool = False, output_file=None, quality_threshold: int = 30, workers: int = 1,
               verbose=False, fraction_low=None, fraction_high=None) -> EndMotifFreqs:
    """Genome-wide end-motif frequencies: counts summed over the 1 Mb windows of every reference
    contig (a fragment crossing a window boundary counts in both, as in the reference)."""
    t0 = time.time()
    min_length, max_length = resolve_motif_aliases(min_length, max_length, fraction_low, fraction_high)
    _clamp_min(min_length, k, "min_length")
    spec = _spec(k, both_strands, negative_strand)
    with ReferenceGenome(refseq_file) as ref:
        windows = genome_windows(ref.chroms)
    hist = region_histograms(input_file, refseq_file, windows, spec, quality_threshold, workers)
    total = hist.sum(axis=0, dtype=np.float64)
    results = EndMotifFreqs(zip(gen_kmers(k), total / np.sum(total)), k, quality_threshold)
    write_motif_freqs(results, output_file)
    if verbose:
        stdout.write(f"end_motifs took {time.time() - t0} seconds to run\n")
    return results


def interval_end_motifs(input_file, refseq_file, intervals, k: int = 4, min_length=50, max_length=None,
                        both_strands: bool = True, negative_strand: bool = False, output_file=None,
                        quality_threshold: int = 30, workers: int = 1, verbose=False, fraction_low=None,
                        fraction_high=None) -> EndMotifsIntervals:
    """End-motif counts for each interval of a BED file / list of ``(chrom, start, stop, name)``."""
    t0 = time.time()
    min_length, max_length = resolve_motif_aliases(min_length, max_length, fraction_low, fraction_high)
    _clamp_min(min_length, k, "min_length")
    spec = _spec(k, both_strands, negative_strand)
    tuples = parse_intervals_arg(intervals)
    hist = region_histograms(input_file, refseq_file, tuples, spec, quality_threshold, workers)
    kmers = gen_kmers(k)
    results = EndMotifsIntervals([(iv, dict(zip(kmers, row.tolist()))) for iv, row in zip(tuples, hist)], k,
                                 quality_threshold)
    write_motif_freqs(results, output_file)
    if verbose:
        stdout.write(f"end_motifs took {time.time() - t0} seconds to run\n")
    return results


def _cli_mds(file_path: str, sep: str = "\t", header: int = 0) -> None:
    stdout.write(f"{EndMotifFreqs.from_file(file_path, 30, sep, header).motif_diversity_score()}\n")


def _cli_regional_mds(file_path: str, file_out: str, sep: str = ",", header: int = 0,
                      miller_madow: bool = False) -> None:
    EndMotifsIntervals.from_file(file_path, 30, sep, header).mds_bed(file_out, miller_madow=miller_madow)
