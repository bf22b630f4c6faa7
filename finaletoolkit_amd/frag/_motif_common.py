"""
Shared pieces of the end-motif and breakpoint-motif features
(``src/finaletoolkit/frag/_motif_common.py`` of the reference): the two result
containers with their readers/writers and motif-diversity score, and the driver
that turns a list of regions into per-region k-mer histograms.

The reference runs one worker call per region, each re-opening the alignment
file and querying the reference genome twice per fragment.  Here the regions of
a contig are counted in ONE ``ftk_motif_counts`` launch against the contig's
fragments and reference image, both resident in HBM.
"""
from __future__ import annotations

import gzip
import itertools
import warnings
from pathlib import Path
from sys import stdin, stdout

import numpy as np

from .. import sharding
from ..reference import ReferenceGenome
from ..source import get_engine, open_source
from ..utils import gen_kmers  # noqa: F401  (the reference keeps it in utils; re-exported here for the motif modules)

MIN_QUALITY = 20           # Jiang et al. (2020); _motif_common.py:30 of the reference
_WINDOW_SIZE = 1_000_000   # genome-wide features are summed over 1 Mb windows (:33)


def normalized_shannon_mds(freq, k: int, miller_madow: bool = False, n=None) -> float:
    """Entropy of the motif frequencies over log(4**k) (_motif_common.py:38-95); optional
    Miller-Madow term (occupied - 1) / (2n)."""
    f = np.asarray(freq, dtype=np.float64)
    logs = np.zeros_like(f)
    np.log(f, out=logs, where=(f != 0))
    h = -np.sum(f * logs)
    if miller_madow:
        if n is None:
            raise ValueError("n is required when miller_madow is True.")
        if not n > 0:
            return float("nan")
        h = h + (int(np.count_nonzero(np.nan_to_num(f))) - 1) / (2 * n)
    return float(h / np.log(4 ** k))


def resolve_motif_aliases(min_length, max_length, fraction_low, fraction_high):
    """Deprecated ``fraction_low`` / ``fraction_high`` (_motif_common.py:98-139)."""
    for old, new, name_old, name_new in ((fraction_low, min_length, "fraction_low", "min_length"),
                                         (fraction_high, max_length, "fraction_high", "max_length")):
        if old is not None:
            warnings.warn(f"{name_old} is deprecated. Use {name_new} instead.", category=DeprecationWarning,
                          stacklevel=3)
            if new is not None:
                raise ValueError(f"{name_old} and {name_new} cannot both be specified")
    if fraction_low is not None:
        min_length = fraction_low
    if fraction_high is not None:
        max_length = fraction_high
    return min_length, max_length


def _open_text(path, mode="r"):
    """(handle, close?) for a path, ``-`` (stdio) or a ``gz`` file."""
    if str(path) == "-":
        return (stdin if mode == "r" else stdout), False
    if mode == "r" and str(path).endswith("gz"):
        return gzip.open(path, "rt"), True
    return open(path, mode), True


class MotifFreqs:
    """Genome-wide k-mer frequencies (``_MotifFreqs``, _motif_common.py:142-268)."""

    def __init__(self, kmer_frequencies, k: int, quality_threshold: int = MIN_QUALITY):
        self.freq_dict = dict(kmer_frequencies)
        self.k = k
        self.quality_threshold = quality_threshold
        if any(len(kmer) != k for kmer in self.freq_dict):
            raise ValueError("kmer_frequencies contains a kmer with length not equal to k.")

    def __iter__(self):
        return iter(self.freq_dict.items())

    def __len__(self):
        return len(self.freq_dict)

    def __str__(self):
        return "".join(f"{kmer}: {freq}\n" for kmer, freq in self)

    def kmers(self) -> list:
        return list(self.freq_dict)

    def frequencies(self) -> list:
        return list(self.freq_dict.values())

    def freq(self, kmer: str) -> float:
        return self.freq_dict[kmer]

    def to_tsv(self, output_file, sep: str = "\t") -> None:
        if not isinstance(output_file, (str, Path)):
            raise TypeError("output_file must be a string or path.")
        out, close = _open_text(output_file, "w")
        try:
            for kmer, freq in self:
                out.write(f"{kmer}{sep}{freq}\n")
        finally:
            if close:
                out.close()

    def motif_diversity_score(self) -> float:
        return normalized_shannon_mds(np.array(self.frequencies()), self.k)

    @classmethod
    def from_file(cls, file_path, quality_threshold: int, sep: str = "\t", header: int = 0):
        fh, close = _open_text(file_path)
        try:
            for _ in range(header):
                fh.readline()
            lines = fh.readlines()
        finally:
            if close:
                fh.close()
        k = len(lines[header].split(sep)[0])  # as the reference: inferred from entry `header` of the body
        pairs = []
        for line in lines:
            fields = line.split(sep)
            if len(fields) != 2:
                break
            pairs.append((fields[0], float(fields[1])))
            if len(fields[0]) != k:
                raise RuntimeError("File contains k-mers of inconsistent length.")
        if len(pairs) != 4 ** k:
            raise RuntimeError(f"File contains {len(pairs)} {k}-mers instead of the expected {4 ** k} {k}-mers.")
        return cls(pairs, k, quality_threshold)


class MotifsIntervals:
    """k-mer counts per interval (``_MotifsIntervals``, _motif_common.py:271-517)."""

    def __init__(self, intervals, k: int, quality_threshold: int = MIN_QUALITY, total_counts=None):
        self.intervals = intervals
        self.k = k
        self.quality_threshold = quality_threshold
        self.total_counts = total_counts
        if any(len(freqs) != 4 ** k for _, freqs in intervals):
            raise ValueError("bins contains results for kmer with length not equal to k.")
        if total_counts is not None and len(total_counts) != len(intervals):
            raise ValueError("total_counts must have one entry per interval.")

    def __iter__(self):
        return iter(self.intervals)

    def __len__(self):
        return len(self.intervals)

    def __str__(self):
        return f"{type(self).__name__} over {len(self.intervals)} intervals."

    @classmethod
    def from_file(cls, file_path: str, quality_threshold: int, sep: str = ",", header: int = 0):
        fh, close = _open_text(file_path)
        try:
            for _ in range(header):
                fh.readline()
            lines = fh.readlines()
        finally:
            if close:
                fh.close()
        kmers = lines[0].split(sep)[5:]
        k = round(np.log(len(kmers)) / np.log(4))
        assert 4 ** k == len(kmers), f"k={k} but should be {len(kmers)}."
        intervals, totals = [], []
        for line in lines[1:]:
            contig, start, stop, name, count, *vals = line.split(sep)
            intervals.append(((contig, int(start), int(stop), name), dict(zip(kmers, map(float, vals)))))
            totals.append(float(count))
        return cls(intervals, k, quality_threshold, totals)

    def freq(self, kmer: str):
        return dict((*interval, freqs[kmer]) for interval, freqs in self.intervals)

    def motif_diversity_score(self, miller_madow: bool = False):
        scores = []
        for i, (interval, freqs) in enumerate(self.intervals):
            counts = np.array(list(freqs.values()))
            total = np.sum(counts)
            n = self.total_counts[i] if self.total_counts is not None else total
            with np.errstate(invalid="ignore", divide="ignore"):
                scores.append((interval, normalized_shannon_mds(counts / total, self.k, miller_madow, n)))
        return scores

    def mds_bed(self, output_file, sep: str = "\t", miller_madow: bool = False) -> None:
        with open(output_file, "w") as out:
            for (contig, start, stop, name), score in self.motif_diversity_score(miller_madow):
                out.write(sep.join([contig, str(start), str(stop), name, str(score)]) + "\n")

    def to_tsv(self, output_file, calc_freq: bool = True, sep: str = "\t") -> None:
        if not isinstance(output_file, (str, Path)):
            raise TypeError("output_file must be a string or path.")
        out, close = _open_text(output_file, "w")
        try:
            out.write(sep.join(["contig", "start", "stop", "name", "count", *gen_kmers(self.k)]) + "\n")
            for interval, freqs in self.intervals:
                total = sum(freqs.values())
                if calc_freq:
                    vals = [f"{v / total:.6f}" if total != 0 else "NaN" for v in freqs.values()]
                else:
                    vals = [str(v) for v in freqs.values()]
                out.write(sep.join([interval[0], str(interval[1]), str(interval[2]), str(interval[3]), str(total),
                                    *vals]) + "\n")
        finally:
            if close:
                out.close()

    def _one_kmer(self, kmer, output_file, calc_freq, sep, with_name):
        if not isinstance(output_file, (str, Path)):
            raise TypeError("output_file must be a string.")
        out, close = _open_text(output_file, "w")
        try:
            for interval, freqs in self.intervals:
                total = sum(freqs.values())
                if calc_freq:
                    val = f"{freqs[kmer] / total:.6f}" if total != 0 else "NaN"
                else:
                    val = freqs[kmer]
                row = [interval[0], str(interval[1]), str(interval[2])] + ([interval[3]] if with_name else [])
                out.write(sep.join(row + [val]) + "\n")   # like the reference, a raw count must be a str here
        finally:
            if close:
                out.close()

    def to_bedgraph(self, kmer, output_file, calc_freq: bool = True, sep: str = "\t") -> None:
        self._one_kmer(kmer, output_file, calc_freq, sep, False)

    def to_bed(self, kmer, output_file, calc_freq: bool = True, sep: str = "\t") -> None:
        self._one_kmer(kmer, output_file, calc_freq, sep, True)


# ---------------------------------------------------------------------------------------
# drivers
# ---------------------------------------------------------------------------------------
def genome_windows(chroms: dict) -> list[tuple[str, int, int]]:
    """1 Mb tiling of every reference contig plus its tail window (_motif_common.py:523-579)."""
    out = []
    for chrom, length in chroms.items():
        out += [(chrom, s, s + _WINDOW_SIZE) for s in range(0, length - _WINDOW_SIZE, _WINDOW_SIZE)]
        out.append((chrom, length - length % _WINDOW_SIZE, length))
    return out


def parse_intervals_arg(intervals):
    """BED path or list of tuples -> ``(chrom, start, stop, name)`` (_motif_common.py:614-632)."""
    if type(intervals) is str:
        with open(intervals) as fh:
            rows = [line.split() for line in fh.readlines()]
        return [(r[0], int(r[1]), int(r[2]), r[3] if len(r) > 3 else ".") for r in rows]
    if isinstance(intervals, list):
        return intervals
    raise TypeError("Intervals should be string or list.")


def region_histograms(input_file, refseq_file, regions, spec: dict, quality_threshold: int, workers: int = 1):
    """uint32 array ``[len(regions), 4**k]`` of motif counts, regions = ``(contig, start, stop, ...)``.

    ``spec``: k, fwd_offset, rev_offset, both_strands, negative_strand, guard, rev_oob_is_error.
    Regions are grouped per contig (one launch each) and returned in input order.  Contigs missing
    from the input file or the reference give zero rows: the reference's per-fragment ``except
    ValueError: continue`` has the same effect for a contig the genome lacks."""
    k = spec["k"]
    out = np.zeros((len(regions), 4 ** k), np.uint32)
    if not regions:
        return out
    src = open_source(input_file, workers, warn_bed6=True)
    eng = get_engine()
    # The reference hands the regions to Pool(workers) (_motif_common.py:635-685); here they are cut into equal-cost runs
    # over the ranks of the process group (sharding.IntervalPlan: whole contigs, a region of the contig a cut falls
    # into; a k-mer count depends only on its own contig's fragments and reference), a rank decodes / uploads / counts
    # only its share, and one all-gather of the count rows gives every rank the whole table.
    lim = 2 ** 31 - 1
    plan = sharding.IntervalPlan([r[0] for r in regions], np.clip([int(r[1]) for r in regions], -lim - 1, lim),
                                 np.clip([int(r[2]) for r in regions], -lim - 1, lim))
    local, err = {}, None
    try:
        with ReferenceGenome(refseq_file) as ref:
            for unit in plan.mine:
                contig = unit[0]
                idx = plan.intervals(unit)
                local[unit] = np.zeros((len(idx), 4 ** k), np.int64)
                if not src.has(contig) or contig not in ref.chroms:
                    continue
                rid = ref.device_image(eng, contig)
                # (a fragment end reads up to k bases beyond the fragment: a region's table holds whole fragments)
                key = plan.unit_key(src, unit, 1)
                try:
                    counts, _, errs = eng.motif_counts(key, rid, plan.starts[idx], plan.stops[idx], k, spec["fwd_offset"],
                                                       spec["rev_offset"], spec["both_strands"], spec["negative_strand"],
                                                       spec["guard"], spec["rev_oob_is_error"], quality_threshold,
                                                       bam=src.is_bam)
                finally:
                    plan.release(src, key)
                if errs.any():
                    j = int(idx[int(np.flatnonzero(errs)[0])])
                    raise RuntimeError(
                        f"Error querying sequence at the 3' end of a fragment in {contig}:{regions[j][1]}-"
                        f"{regions[j][2]}. Chrom length: {ref.chroms.get(contig, 'unknown')}. Please verify that the "
                        "reference file matches the fragment file.")
                local[unit][:] = counts
    except Exception as e:  # noqa: BLE001 - every rank learns of it below
        err = e
    sharding.agree(err)
    out[:] = plan.gather(local, 4 ** k)
    return out


def write_motif_freqs(results, output_file) -> None:
    """TSV, or CSV for a ``.csv`` suffix (_motif_common.py:688-697)."""
    if output_file is None or not sharding.is_writer():  # every rank holds the result; rank 0 alone writes it
        return
    if output_file.endswith(".csv"):
        results.to_tsv(output_file, sep=",")
    else:
        results.to_tsv(output_file)
