"""
DELFI short/long fragment features per genomic bin -- ``delfi`` with the
reference's signature, bin filtering, arm gating, ratio, positional no-coverage
drop, merge and writers (``src/finaletoolkit/frag/_delfi.py:129-511``).

The per-fragment loop of ``_delfi_single_window`` (``:443-472``) runs on the
MI355X: one ``ftk_delfi_counts`` launch per contig over all of its ungated
bins.  Window gating (``:423-428``), GC fraction (``:476-490``), ratio and the
5 Mb merge are O(n_bins) host arithmetic.
"""
from __future__ import annotations

import time
import warnings
from collections import defaultdict
from sys import stderr, stdout
from typing import Union

import numpy as np
import pandas

from ..genome.gaps import GenomeGaps
from ..reference import ReferenceGenome
from ..source import get_engine, open_source
from .. import sharding
from ..utils import chrom_sizes_to_list, overlaps
from ._delfi_gc_correct import delfi_gc_correct
from ._delfi_merge_bins import delfi_merge_bins

__all__ = ["delfi", "trim_coverage"]


def trim_coverage(window_data: np.ndarray, trim_percentile: int = 10):
    """frag/_delfi.py:32-45: blank the lowest ``trim_percentile`` % of bins by fragment count."""
    threshold = np.percentile(window_data["num_frags"], trim_percentile)
    trimmed = window_data.copy()
    low = window_data["num_frags"] < threshold
    trimmed["short"][low] = np.nan
    trimmed["long"][low] = np.nan
    trimmed["gc"][low] = np.nan
    trimmed["num_frags"][low] = 0
    return trimmed


def _load_blacklist_indexed(blacklist_file):
    """contig -> (sorted starts, stops)  (frag/_delfi.py:85-107)."""
    if blacklist_file is None:
        return {}
    by_contig = defaultdict(list)
    with open(blacklist_file) as fh:
        for line in fh:
            parts = line.split()
            if len(parts) < 3:
                continue
            by_contig[parts[0]].append((int(parts[1]), int(parts[2])))
    out = {}
    for contig, regions in by_contig.items():
        regions.sort()
        out[contig] = (np.array([r[0] for r in regions], dtype=np.int64),
                       np.array([r[1] for r in regions], dtype=np.int64))
    return out


def _resolve_gaps(gap_file):
    if gap_file is None:
        return None
    if isinstance(gap_file, str):
        return GenomeGaps(gap_file)
    if isinstance(gap_file, GenomeGaps):
        return gap_file
    raise TypeError(f"{type(gap_file)} is not accepted type for gap_file")


def _valid_interval(chroms: dict, contig, start, stop) -> bool:
    """utils/validation.py:111-183 for a dict of contig lengths."""
    if contig not in chroms:
        return False
    length = chroms[contig]
    if start < 0 or start >= length or stop < 0 or stop > length or start >= stop:
        return False
    return True


def _gate_windows(contig, starts, stops, contig_gaps):
    """Window-level gating of frag/_delfi.py:416-428 (host, every rank): arm label per bin and which bins go
    to the device."""
    n = len(starts)
    arms = [contig] * n
    live = np.ones(n, dtype=bool)
    if contig_gaps is not None:
        for i in range(n):
            if contig_gaps.in_tcmere(starts[i], stops[i]):
                arms[i], live[i] = "NOARM", False
                continue
            arm = contig_gaps.get_arm(starts[i], stops[i])
            arms[i] = arm
            if arm == "NOARM":
                live[i] = False
    return arms, live


def _contig_counts(src, eng, ref, contig, starts, stops, live, ok, contig_gaps, blacklist, quality_threshold):
    """Device part of one contig (the rank that owns it): ``[n_live, 4]`` int64 rows
    ``(short, long, num_frags, num_gc)`` of its live bins -- one ``ftk_delfi_counts`` launch and one GC-count
    launch (frag/_delfi.py:443-490)."""
    idx = np.nonzero(live)[0]
    out = np.zeros((len(idx), 4), np.int64)
    if len(idx):
        bl = blacklist.get(contig)
        sh, lg, nf = eng.delfi_counts(
            src.require(contig), starts[idx].astype(np.int32), stops[idx].astype(np.int32), quality_threshold,
            None if bl is None else bl[0], None if bl is None else bl[1],
            None if contig_gaps is None else contig_gaps.as_kernel_constants())
        out[:, 0], out[:, 1], out[:, 2] = sh, lg, nf
        if ok.any():
            out[ok, 3] = ref.gc_counts(eng, contig, starts[idx][ok], stops[idx][ok])
    return out


def _contig_rows(contig, starts, stops, arms, live, ok, counts):
    """Rows ``(contig, start, stop, arm, short, long, gc, num_frags)`` of one contig's bins in bin order
    (frag/_delfi.py:404-511), from the gathered device counts (host, every rank)."""
    rows = [None] * len(starts)
    for k, i in enumerate(np.nonzero(live)[0]):
        ws, we = int(starts[i]), int(stops[i])
        if not ok[k]:
            warnings.warn(f"Invalid interval {contig}:{ws}-{we} for reference. Skipping GC calculation.")
        sh, lg, nf, num_gc = (int(v) for v in counts[k])
        gc = num_gc / (we - ws) if nf > 0 else np.nan
        rows[i] = (contig, ws, we, arms[i], sh, lg, gc, nf)
    for i in np.nonzero(~live)[0]:
        rows[i] = (contig, int(starts[i]), int(stops[i]), "NOARM", np.nan, np.nan, np.nan, 0)
    return rows


def _contig_windows(src, eng, ref, contig, starts, stops, contig_gaps, blacklist, quality_threshold):
    """One contig start to finish on this GPU: gate, count, assemble (what a 1-rank ``delfi`` does per contig)."""
    arms, live = _gate_windows(contig, starts, stops, contig_gaps)
    ok = np.array([_valid_interval(ref.chroms, contig, int(starts[i]), int(stops[i])) for i in np.nonzero(live)[0]],
                  dtype=bool)
    counts = _contig_counts(src, eng, ref, contig, starts, stops, live, ok, contig_gaps, blacklist, quality_threshold)
    return _contig_rows(contig, starts, stops, arms, live, ok, counts)


def delfi(input_file: str, chrom_sizes: str, bins_file: str, reference_file: str, blacklist_file: str = None,
          gap_file: Union[str, GenomeGaps] = None, output_file: str = None, no_gc_correct: bool = False,
          gc_correct: bool | None = None, remove_nocov: bool = True, merge_bins: bool = True,
          window_size: int = 5000000, quality_threshold: int = 30, workers: int = 1,
          verbose: Union[int, bool] = False) -> pandas.DataFrame:
    """DELFI features (Cristiano et al., 2019); returns the result frame."""
    if verbose:
        t0 = time.time()
        stderr.write(f"delfi: {input_file} bins {bins_file}\n")
    contigs = chrom_sizes_to_list(chrom_sizes)
    if gc_correct is None:
        gc_correct = not no_gc_correct
    else:
        warnings.warn("Warning: gc_correct is deprecated and may be removed in future releases. "
                      "Use no_gc_correct instead")
    gaps = _resolve_gaps(gap_file)

    bins = pandas.read_csv(bins_file, names=["contig", "start", "stop"], usecols=[0, 1, 2],
                           dtype={"contig": str, "start": np.int32, "stop": np.int32}, delimiter="\t", comment="#")
    if gaps is not None:
        in_gap = overlaps(bins["contig"].to_numpy(), bins["start"].to_numpy(), bins["stop"].to_numpy(),
                          gaps.gaps["contig"], gaps.gaps["start"], gaps.gaps["stop"])
        gapless_bins = bins.loc[~in_gap]
    else:
        gapless_bins = bins

    blacklist = _load_blacklist_indexed(blacklist_file)
    contig_gaps = {}
    if gaps is not None:
        for contig, _size in contigs:
            contig_gaps[contig] = gaps.get_contig_gaps(contig)

    src = open_source(input_file, workers)
    eng = get_engine()
    # The reference fans the bins out over Pool(workers) (frag/_delfi.py:289-300).  Here the fan-out is one
    # rank per GPU: contigs are dealt to the ranks of the initialised process group (LPT on bin counts), a
    # rank decodes and counts only its own contigs, and ONE all-gather of the per-bin
    # (short, long, num_frags, num_gc) vector gives every rank the whole table -- the frame returned is
    # the same on all ranks and equal to the single-GPU one.
    rank, world = sharding.rank_world()
    plan = []  # (contig, starts, stops, arms, live, ok) in chrom.sizes order, bins in file order (:269-283)
    with ReferenceGenome(reference_file) as ref:
        for contig, _size in contigs:
            sel = gapless_bins.loc[gapless_bins["contig"] == contig]
            if sel.shape[0] == 0:
                continue
            starts = sel["start"].to_numpy().astype(np.int64)
            stops = sel["stop"].to_numpy().astype(np.int64)
            arms, live = _gate_windows(contig, starts, stops, contig_gaps.get(contig) if gaps is not None else None)
            ok = np.array([_valid_interval(ref.chroms, contig, int(starts[i]), int(stops[i]))
                           for i in np.nonzero(live)[0]], dtype=bool)
            plan.append((contig, starts, stops, arms, live, ok))
        names = [p[0] for p in plan]
        weights = {p[0]: float(len(p[1])) for p in plan}
        owner = sharding.lpt_assign(weights, world)
        if world == 1 and 2 * len(names) >= len(src.contigs):
            src.load_all()  # most of the file is needed: one streaming pass instead of one index seek per contig
        local = {}
        for contig, starts, stops, arms, live, ok in plan:
            if owner[contig] == rank:
                local[contig] = _contig_counts(src, eng, ref, contig, starts, stops, live, ok,
                                               contig_gaps.get(contig) if gaps is not None else None, blacklist,
                                               quality_threshold)
    n_live = {p[0]: int(p[4].sum()) for p in plan}
    counts = sharding.gather_bin_vectors(local, names, n_live, weights, k=4)
    windows = []
    for contig, starts, stops, arms, live, ok in plan:
        windows += _contig_rows(contig, starts, stops, arms, live, ok, counts[contig])

    window_df = pandas.DataFrame(windows, columns=["contig", "start", "stop", "arm", "short", "long", "gc",
                                                   "num_frags"])
    trimmed = window_df.loc[window_df["arm"] != "NOARM", :].copy()
    trimmed["ratio"] = np.where(trimmed["long"] == 0, np.nan, trimmed["short"] / trimmed["long"])

    if remove_nocov:  # the two hg19 no-coverage windows, by position (:333-340)
        pos = np.arange(trimmed.shape[0])
        final = trimmed.loc[np.logical_and(pos != 8779, pos != 13664)].reset_index()
    else:
        final = trimmed
    if gc_correct:
        final = delfi_gc_correct(final, 0.75, 8, verbose)
    if merge_bins:
        final = delfi_merge_bins(final, gc_correct, verbose=verbose)
    if output_file is not None and rank == 0:  # every rank holds the same frame; one of them writes it
        _write_delfi(final, output_file)
    if verbose:
        stderr.write(f"{sum(w[7] for w in windows)} fragments included.\n")
        stderr.write(f"delfi took {time.time() - t0} s to complete\n")
    return final


def _write_delfi(final_bins: pandas.DataFrame, output_file: str) -> None:
    """BED/TSV/CSV/gz or stdout (frag/_delfi.py:384-401)."""
    renamed = final_bins.rename(columns={"contig": "#contig"})
    if output_file.endswith(".bed") or output_file.endswith(".tsv"):
        renamed.to_csv(output_file, sep="\t", index=False)
    elif output_file.endswith(".csv"):
        final_bins.to_csv(output_file, sep=",", index=False)
    elif output_file.endswith(".bed.gz"):
        renamed.to_csv(output_file, sep="\t", index=False, encoding="gzip")
    elif output_file == "-":
        for window in final_bins.itertuples(index=False, name=None):
            stdout.write("\t".join(str(field) for field in window) + "\n")
    else:
        raise ValueError("Invalid file type! Only .bed, .bed.gz, and .tsv suffixes allowed.")
