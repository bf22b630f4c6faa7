"""
Host-side helpers of the hot path with the reference's names and semantics
(``src/finaletoolkit/utils/utils.py`` and ``utils/_frag_generator.py``):
chrom.sizes / BED readers, ``overlaps``, and the fragment stream
(``frag_generator`` / ``frag_array``), which here is a device-side ordered
selection (``ftk_frag_select``) instead of a per-row Python predicate.
"""
from __future__ import annotations

import itertools
from pathlib import Path
from typing import Generator, Tuple

import numpy as np

from .exceptions import InvalidInputError
from .source import get_engine, open_source

__all__ = ["chrom_sizes_to_list", "chrom_sizes_to_dict", "get_intervals", "overlaps", "frags_in_region", "frag_generator",
           "frag_array", "agg_bw", "gen_kmers", "reverse_complement", "validate_compatible_contigs", "valid_interval",
           "_none_eq", "_none_geq", "_none_leq"]

FragTuple = Tuple[str, int, int, int, bool]


def chrom_sizes_to_list(chrom_sizes_file) -> list[tuple[str, int]]:
    """utils/utils.py:53-73."""
    out = []
    with open(chrom_sizes_file, "r") as fh:
        for line in fh:
            if line != "\n":
                chrom, size = line.strip().split("\t")
                out.append((chrom, int(size)))
    return out


def chrom_sizes_to_dict(chrom_sizes_file) -> dict[str, int]:
    """utils/utils.py:76-94."""
    return dict(chrom_sizes_to_list(chrom_sizes_file))


def get_intervals(interval_file) -> list[tuple[str, int, int, str]]:
    """BED reader (utils/utils.py:310-343): skips ``#``/``track``/``browser``/blank
    lines and rows with < 3 columns; missing name -> ``'.'``."""
    intervals = []
    with open(interval_file, "r") as bed:
        for line in bed:
            if line.startswith(("#", "track", "browser")) or not line.strip():
                continue
            parts = line.strip().split("\t")
            if len(parts) < 3:
                continue
            intervals.append((parts[0], int(parts[1]), int(parts[2]), parts[3] if len(parts) > 3 else "."))
    return intervals


def overlaps(contigs_1, starts_1, stops_1, contigs_2, starts_2, stops_2):
    """Does each interval of set 1 overlap any interval of set 2 on the same
    contig?  (utils/utils.py:346-382; grouped by contig instead of an
    n1 x n2 broadcast.)"""
    contigs_1 = np.asarray(contigs_1)
    starts_1 = np.asarray(starts_1)
    stops_1 = np.asarray(stops_1)
    contigs_2 = np.asarray(contigs_2)
    starts_2 = np.asarray(starts_2)
    stops_2 = np.asarray(stops_2)
    out = np.zeros(contigs_1.shape[0], dtype=bool)
    for c in np.unique(contigs_1):
        m1 = contigs_1 == c
        m2 = contigs_2 == c
        if not m2.any():
            continue
        s1 = starts_1[m1][:, None]
        e1 = stops_1[m1][:, None]
        out[m1] = np.any((s1 < stops_2[m2][None]) & (e1 > starts_2[m2][None]), axis=1)
    return out


def gen_kmers(k: int, bases: str = "ACGT") -> list[str]:
    """All ``len(bases)**k`` k-mers in lexicographic order (utils/utils.py:388-410)."""
    if k < 0:
        raise ValueError("k must be non-negative")
    return ["".join(t) for t in itertools.product(bases, repeat=k)]


def frags_in_region(frag_array, start: int, stop: int):
    """Rows of a ``frag_array`` result with ``start < stop_`` and ``stop >= start_`` (utils/utils.py:160-183;
    note the inclusive lower test)."""
    keep = (frag_array["start"] < stop) & (frag_array["stop"] >= start)
    return frag_array[keep]


def _check_policy(intersect_policy: str):
    if intersect_policy not in ("midpoint", "any"):
        raise InvalidInputError(f"{intersect_policy} is not a valid policy")


def _check_region(contig, start, stop):
    # utils/_frag_generator.py:105-110
    if contig is None and not (start is None and stop is None):
        if not (start == 0 and stop is None):
            raise InvalidInputError("contig should be specified if start or stop given.")


def _region_contigs(src, contig):
    """Contigs a fetch(contig, ...) touches, in file order, with their bounds
    semantics: ``contig=None`` iterates the whole file and pysam ignores
    start/stop (io/alignment.py:245,273-279)."""
    if contig is None:
        src.load_all()
        return [c for c in src.contigs if c in src.loaded], True
    return [contig], False


def frag_generator(input_file, contig, quality_threshold: int = 30, start=None, stop=None, min_length=None,
                   max_length=None, intersect_policy: str = "midpoint", verbose=False,
                   reference_file=None) -> Generator[FragTuple, None, None]:
    """Stream ``(contig, start, stop, mapq, is_forward)`` of the fragments
    passing the shared predicate (utils/_frag_generator.py:58-141)."""
    _check_policy(intersect_policy)
    _check_region(contig, start, stop)
    src = open_source(input_file)
    src.check_fetch(contig, start, stop)
    eng = get_engine()
    names, whole = _region_contigs(src, contig)
    for c in names:
        key = src.require(c)
        s, e, q, st = eng.frag_select(key, None if whole else start, None if whole else stop, quality_threshold,
                                      min_length, max_length, intersect_policy)
        for i in range(len(s)):
            yield (c, int(s[i]), int(e[i]), int(q[i]), bool(st[i]))


def frag_array(input_file, contig: str, quality_threshold: int = 30, start=None, stop=None, min_length=None,
               max_length=None, intersect_policy: str = "midpoint", verbose: bool = False, reference_file=None):
    """Structured ``[('start','i8'),('stop','i8'),('strand','?')]`` array of the
    passing fragments (utils/utils.py:186-255)."""
    _check_policy(intersect_policy)
    _check_region(contig, start, stop)
    src = open_source(input_file)
    src.check_fetch(contig, start, stop)
    eng = get_engine()
    names, whole = _region_contigs(src, contig)
    parts = []
    for c in names:
        s, e, _, st = eng.frag_select(src.require(c) if whole else src.require_interval(c, start, stop, 1), None if whole else start, None if whole else stop,
                                      quality_threshold, min_length, max_length, intersect_policy)
        a = np.zeros(len(s), dtype=[("start", "i8"), ("stop", "i8"), ("strand", "?")])
        a["start"], a["stop"], a["strand"] = s, e, st.astype(bool)
        parts.append(a)
    if not parts:
        return np.zeros(0, dtype=[("start", "i8"), ("stop", "i8"), ("strand", "?")])
    return np.concatenate(parts)


def agg_bw(input_file, interval_file, output_file, median_window_size: int = 1, mean: bool = False,
           verbose: bool = False) -> np.ndarray:
    """Aggregate a bigWig signal across strand-oriented intervals (reference: ``utils/_agg_bw.py:18-146``):
    every interval's per-base values (NaN -> 0) are trimmed by the upstream median filter's window
    (``values[w // 2 : -w // 2]`` - with ``w = 0`` that slice is empty and every interval is skipped, as in the
    reference), flipped for ``-`` strand intervals, and summed; ``mean`` divides by the intervals added.  A host
    utility downstream of ``adjust_wps`` - file reading and a few vector additions, no kernel."""
    import gzip
    import time
    from sys import stderr

    from .bigwig import BigWigFile
    t0 = time.time()
    if not (str(interval_file).endswith(".bed") or str(interval_file).endswith(".bed.gz")):
        raise ValueError("Invalid filetype for interval_file.")
    intervals = []
    opener = gzip.open if str(interval_file).endswith(".gz") else open
    with opener(interval_file, "rt") as fh:
        for line in fh:
            f = line.split("\t")
            intervals.append((f[0], int(f[1]), int(f[2]), f[5].strip()))
    with BigWigFile(str(input_file)) as bw:
        interval_size = intervals[0][2] - intervals[0][1] - median_window_size
        agg = np.zeros(interval_size, dtype=np.int64)
        added = 0
        for contig, start, stop, strand in intervals:
            try:
                values = np.nan_to_num(bw.values(contig, start, stop), nan=0)
            except RuntimeError as e:
                print(e)
                continue
            trimmed = values[median_window_size // 2: -median_window_size // 2]
            if trimmed.shape[0] != interval_size:
                print(f"Trimmed size {trimmed.shape[0]} for {contig}:{start}-{stop} is not equal to "
                      f"interval size {interval_size}. Skipping.")
                continue
            if strand == "+":
                agg = agg + trimmed
                added += 1
            elif strand == "-":
                agg = agg + np.flip(trimmed)
                added += 1
            elif verbose:
                stderr.write("A segment without strand was encountered. Skipping.")
    if mean:
        agg = agg / added
    if not str(output_file).endswith("wig"):
        raise ValueError("The output_file is an unaccepted type. Must be a wiggle file ending in .wig")
    with open(output_file, "wt") as out:
        out.write(f"fixedStep\tchrom=.\tstart={-interval_size // 2}\tstep={1}\tspan={interval_size}\n")
        for score in agg:
            out.write(f"{score}\n")
    if verbose:
        stderr.write(f"Aggregating bigWig took {time.time() - t0} s to complete\n")
    return agg


# ---- small pure helpers of the reference's utility layer (kept so that `finaletoolkit.utils.<name>` resolves) --------
from .validation import valid_interval, validate_compatible_contigs  # noqa: E402

_COMPLEMENT = str.maketrans("ACGTacgt", "TGCATGCA")


def reverse_complement(kmer: str) -> str:
    """utils/utils.py:413-437: the reverse complement of a DNA string - A/C/G/T in either case give the upper-case
    complement, every other character (``N``) stays what it is."""
    return kmer.translate(_COMPLEMENT)[::-1]


def _none_leq(a, b) -> bool:
    """utils/_comparison.py: ``a <= b`` where a missing operand means "unbounded" (true)."""
    return a is None or b is None or a <= b


def _none_geq(a, b) -> bool:
    return a is None or b is None or a >= b


def _none_eq(a, b) -> bool:
    return a is None or b is None or a == b
