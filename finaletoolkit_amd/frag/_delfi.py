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

import threading
import time
import warnings
from collections import defaultdict
from sys import stderr, stdout
from typing import Union

import numpy as np

from .._lazy import LazyModule
from ..genome.gaps import GenomeGaps
from ..reference import ReferenceGenome
from ..engine import Engine
from ..source import EarlyContigs, get_engine, get_side_engine, region_contig, resident_contigs
from .. import sharding
from ..utils import chrom_sizes_to_list, overlaps
from ._delfi_gc_correct import delfi_gc_correct
from ._delfi_merge_bins import delfi_merge_bins

pandas = LazyModule("pandas")  # (imported by the first DELFI call, not by `import finaletoolkit_amd.frag`: 0.2-0.6 s)

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


def _valid_mask(chroms: dict, contig, starts, stops) -> np.ndarray:
    """utils/validation.py:111-183 for a dict of contig lengths, all bins of a contig at once."""
    if contig not in chroms:
        return np.zeros(len(starts), dtype=bool)
    length = chroms[contig]
    return ~((starts < 0) | (starts >= length) | (stops < 0) | (stops > length) | (starts >= stops))


def _gate_windows(contig, starts, stops, contig_gaps):
    """Window-level gating of frag/_delfi.py:416-428 (host, every rank) for all bins of a contig at once: arm
    label per bin (object array) and which bins go to the device.  ``ContigGaps.in_tcmere`` / ``get_arm`` as
    array expressions: inside the centromere or overlapping EVERY telomere -> NOARM; else left of the
    centromere -> ``p`` (NOARM on a short-arm contig), right of it -> ``q``, else NOARM."""
    n = len(starts)
    if contig_gaps is None:
        arms = np.empty(n, dtype=object)
        arms[:] = contig
        return arms, np.ones(n, dtype=bool)
    c0, c1 = contig_gaps.centromere
    tcmere = (stops > c0) & (starts < c1)
    if contig_gaps.telomeres:
        every = np.ones(n, dtype=bool)
        for t0, t1 in contig_gaps.telomeres:
            every &= (stops > t0) & (starts < t1)
        tcmere |= every
    if np.any(~tcmere & (stops < starts)):
        raise ValueError("start must be less than stop")
    left = ~tcmere & (stops < c0)
    p_arm = left & (not contig_gaps.has_short_arm)
    q_arm = ~tcmere & ~left & (starts > c1)
    name = contig.replace("chr", "")
    arms = np.empty(n, dtype=object)
    arms[:] = "NOARM"
    arms[p_arm] = f"{name}p"
    arms[q_arm] = f"{name}q"
    return arms, p_arm | q_arm


def _unit_bins(live, ok, i0=0, i1=None):
    """Indices (into the contig's bins) of the live bins of unit ``[i0, i1)`` and, for each, whether the reference
    covers it (``ok`` is indexed like the contig's live bins)."""
    idx = np.nonzero(live)[0]
    if i0 or i1 is not None:
        part = (idx >= i0) & (idx < (len(live) if i1 is None else i1))
        idx, ok = idx[part], ok[part]
    return idx, ok


class _GCAhead:
    """G + C per bin (frag/_delfi.py:476-490) of every unit this rank owns, counted BESIDE the decode: the counts need
    the reference and the bins, not the fragments, so a worker thread uploads each contig's reference image and counts
    on a second context of the same GPU (``source.get_side_engine``) while the main thread waits for contigs to become
    resident - in round 3 the same 43 ms per genome sat on the consumer thread between the count kernels.  ``jobs``:
    ``[(unit key, contig, starts, stops)]`` in the order the units will be asked for.  With a stand-in device (host
    logic tests) the counts are taken on demand on the caller's thread."""

    def __init__(self, ref, eng, jobs):
        self.ref, self.eng = ref, eng
        self.jobs = {key: (contig, starts, stops) for key, contig, starts, stops in jobs}
        self.done = {key: threading.Event() for key in self.jobs}
        self.out, self.err, self.thread = {}, None, None
        self.stop = False  # set by close(abort=True): the worker gives up between two contigs
        if isinstance(eng, Engine) and jobs:
            self.thread = threading.Thread(target=self._run, args=([j[0] for j in jobs],), name="ftk-delfi-gc", daemon=True)
            self.thread.start()

    def _run(self, order):
        try:
            side = get_side_engine()
            for key in order:
                if self.stop:
                    break
                contig, starts, stops = self.jobs[key]
                self.out[key] = self.ref.gc_counts(side, contig, starts, stops) if len(starts) else np.zeros(0, np.int64)
                self.done[key].set()
        except BaseException as e:  # noqa: BLE001 - re-raised on the caller's thread by get()
            self.err = e
        finally:
            for ev in self.done.values():
                ev.set()

    def get(self, key):
        if self.thread is None:
            contig, starts, stops = self.jobs[key]
            return self.ref.gc_counts(self.eng, contig, starts, stops) if len(starts) else np.zeros(0, np.int64)
        self.done[key].wait()
        if key not in self.out:
            raise self.err if self.err is not None else RuntimeError("the G + C worker stopped early")
        return self.out.pop(key)

    def close(self, abort: bool = False):
        """``abort`` (the caller failed): do not count the remaining contigs' references before the error is reported."""
        if abort:
            self.stop = True
        if self.thread is not None:
            self.thread.join()


def _contig_counts(src, eng, ref, contig, starts, stops, live, ok, contig_gaps, blacklist, quality_threshold,
                   clock=None, key=None, i0=0, i1=None, gc_ahead=None):
    """Device part of one contig (the rank that owns it) - or of the bins ``[i0, i1)`` of it, counted on the table
    ``key`` (a region of the contig: ``FragSource.require_region``): ``[n_live, 4]`` int64 rows
    ``(short, long, num_frags, num_gc)`` of the live bins among them -- one ``ftk_delfi_counts`` launch and the
    G + C counts (frag/_delfi.py:443-490; ``gc_ahead``: taken from the worker that counted them beside the decode)."""
    unit = (contig, i0, len(live) if i1 is None else i1)
    idx, ok = _unit_bins(live, ok, i0, i1)
    out = np.zeros((len(idx), 4), np.int64)
    if len(idx):
        t0 = time.perf_counter()
        if key is None:
            key = src.require(contig)
        t1 = time.perf_counter()
        bl = blacklist.get(contig)
        sh, lg, nf = eng.delfi_counts(
            key, starts[idx].astype(np.int32), stops[idx].astype(np.int32), quality_threshold,
            None if bl is None else bl[0], None if bl is None else bl[1],
            None if contig_gaps is None else contig_gaps.as_kernel_constants())
        out[:, 0], out[:, 1], out[:, 2] = sh, lg, nf
        t2 = time.perf_counter()
        if ok.any():
            out[ok, 3] = (gc_ahead.get(unit) if gc_ahead is not None
                          else ref.gc_counts(eng, contig, starts[idx][ok], stops[idx][ok]))
        if clock is not None:
            clock["decode_wait"] += t1 - t0
            clock["count_kernels"] += t2 - t1
            clock["gc_count"] += time.perf_counter() - t2
    return out


_COLUMNS = ["contig", "start", "stop", "arm", "short", "long", "gc", "num_frags"]


def _contig_columns(contig, starts, stops, arms, live, ok, counts):
    """Columns of one contig's bins in bin order (frag/_delfi.py:404-511), from the gathered device counts (host,
    every rank): NOARM bins carry NaN / NaN / NaN / 0; a live bin's ``gc`` is ``num_gc / (stop - start)`` when it
    holds a fragment, else NaN.  A live bin outside the reference warns and keeps ``num_gc = 0``."""
    n = len(starts)
    idx = np.nonzero(live)[0]
    for k in np.nonzero(~ok)[0]:
        warnings.warn(f"Invalid interval {contig}:{int(starts[idx[k]])}-{int(stops[idx[k]])} for reference. "
                      "Skipping GC calculation.")
    short = np.full(n, np.nan)
    long_ = np.full(n, np.nan)
    gc = np.full(n, np.nan)
    nfrag = np.zeros(n, np.int64)
    if len(idx):
        counts = np.asarray(counts, dtype=np.int64).reshape(-1, 4)
        short[idx], long_[idx], nfrag[idx] = counts[:, 0], counts[:, 1], counts[:, 2]
        width = (stops[idx] - starts[idx]).astype(np.int64)
        with np.errstate(divide="ignore", invalid="ignore"):
            gc[idx] = np.where(counts[:, 2] > 0, counts[:, 3] / width, np.nan)
    names = np.empty(n, dtype=object)
    names[:] = contig
    return names, starts.astype(np.int64), stops.astype(np.int64), arms, short, long_, gc, nfrag


def _window_frame(parts) -> pandas.DataFrame:
    """The frame ``pandas.DataFrame(windows, columns=...)`` builds from the reference's list of row tuples
    (frag/_delfi.py:303-315), from per-contig columns: ``short`` / ``long`` are integer columns unless a NOARM
    row put a NaN into them (then float64, printed as ``12.0``) -- the same inference, without 30 970 tuples."""
    if not parts:
        return pandas.DataFrame([], columns=_COLUMNS)
    cols = [np.concatenate([p[k] for p in parts]) for k in range(8)]
    if not len(cols[0]):
        return pandas.DataFrame([], columns=_COLUMNS)
    for k in (4, 5):
        if not np.isnan(cols[k]).any():
            cols[k] = cols[k].astype(np.int64)
    return pandas.DataFrame(dict(zip(_COLUMNS, cols)), columns=_COLUMNS)


def _contig_windows(src, eng, ref, contig, starts, stops, contig_gaps, blacklist, quality_threshold):
    """One contig start to finish on this GPU: gate, count, assemble -- the rows
    ``(contig, start, stop, arm, short, long, gc, num_frags)`` a 1-rank ``delfi`` produces for it."""
    starts = np.asarray(starts, np.int64)
    stops = np.asarray(stops, np.int64)
    arms, live = _gate_windows(contig, starts, stops, contig_gaps)
    ok = _valid_mask(ref.chroms, contig, starts[live], stops[live])
    counts = _contig_counts(src, eng, ref, contig, starts, stops, live, ok, contig_gaps, blacklist, quality_threshold)
    cols = _contig_columns(contig, starts, stops, arms, live, ok, counts)
    rows = []
    for i in range(len(starts)):
        if live[i]:
            rows.append((contig, int(starts[i]), int(stops[i]), arms[i], int(cols[4][i]), int(cols[5][i]),
                         float(cols[6][i]), int(cols[7][i])))
        else:
            rows.append((contig, int(starts[i]), int(stops[i]), "NOARM", np.nan, np.nan, np.nan, 0))
    return rows


# wall time of the last delfi() call by stage (seconds): what bench.py's `genome_frag_delfi_api` leg reports
LAST_STAGE_S: dict = {}


def delfi(input_file: str, chrom_sizes: str, bins_file: str, reference_file: str, blacklist_file: str = None,
          gap_file: Union[str, GenomeGaps] = None, output_file: str = None, no_gc_correct: bool = False,
          gc_correct: bool | None = None, remove_nocov: bool = True, merge_bins: bool = True,
          window_size: int = 5000000, quality_threshold: int = 30, workers: int = 1,
          verbose: Union[int, bool] = False) -> pandas.DataFrame:
    """DELFI features (Cristiano et al., 2019); returns the result frame."""
    t_begin = time.perf_counter()
    clock = dict(read_inputs=0.0, gate=0.0, decode_wait=0.0, count_kernels=0.0, gc_count=0.0, gather=0.0, frame=0.0,
                 merge=0.0, write=0.0)
    if verbose:
        t0 = time.time()
        stderr.write(f"delfi: {input_file} bins {bins_file}\n")
    # One process: every contig of the file will be wanted (or skipped cheaply), so the decoder starts NOW and works
    # towards chr1 while the side files below are read.  (Several ranks each want their own share: known after the plan.)
    early = None
    if sharding.rank_world()[1] > 1:
        # several ranks: a rank without a usable GPU must say so BEFORE the others enter the first collective
        err = None
        try:
            get_engine()
        except Exception as e:  # noqa: BLE001 - every rank learns of it
            err = e
        sharding.agree(err)
    elif isinstance(get_engine(), Engine):
        try:  # chrom.sizes is a few hundred bytes: the names it lists are all this call can ask the file for
            listed = list(dict.fromkeys(c for c, _ in chrom_sizes_to_list(chrom_sizes)))
        except Exception:  # noqa: BLE001 - _delfi reads it again and raises the error where the reference does
            listed = None
        if listed is not None:
            early = EarlyContigs(input_file, workers, names=listed)
    try:
        return _delfi(early, t_begin, clock, input_file, chrom_sizes, bins_file, reference_file, blacklist_file, gap_file,
                      output_file, no_gc_correct, gc_correct, remove_nocov, merge_bins, window_size, quality_threshold,
                      workers, verbose)
    except BaseException:
        if early is not None:
            early.close()
        raise


def _delfi(early, t_begin, clock, input_file, chrom_sizes, bins_file, reference_file, blacklist_file, gap_file, output_file,
           no_gc_correct, gc_correct, remove_nocov, merge_bins, window_size, quality_threshold, workers, verbose):
    if verbose:
        t0 = time.time()
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
    clock["read_inputs"] = time.perf_counter() - t_begin

    eng = get_engine()
    # The reference fans the bins out over Pool(workers) (frag/_delfi.py:289-300).  Here the fan-out is one
    # rank per GPU: contigs are dealt to the ranks of the initialised process group (LPT on bin counts), a
    # rank decodes and counts only its own contigs, and ONE all-gather of the per-bin
    # (short, long, num_frags, num_gc) vector gives every rank the whole table -- the frame returned is
    # the same on all ranks and equal to the single-GPU one.
    rank, world = sharding.rank_world()
    plan = {}  # contig -> (starts, stops, arms, live, ok), chrom.sizes order, bins in file order (:269-283)
    bin_contig = gapless_bins["contig"].to_numpy()
    bin_start = gapless_bins["start"].to_numpy().astype(np.int64)
    bin_stop = gapless_bins["stop"].to_numpy().astype(np.int64)
    with ReferenceGenome(reference_file) as ref:
        tg = time.perf_counter()
        for contig, _size in contigs:
            if contig in plan:
                continue
            sel = np.nonzero(bin_contig == contig)[0]
            if len(sel) == 0:
                continue
            starts, stops = bin_start[sel], bin_stop[sel]
            arms, live = _gate_windows(contig, starts, stops, contig_gaps.get(contig) if gaps is not None else None)
            plan[contig] = (starts, stops, arms, live, _valid_mask(ref.chroms, contig, starts[live], stops[live]))
        clock["gate"] = time.perf_counter() - tg
        names = list(plan)
        # The ranks' shares: the bins of all contigs laid end to end and cut into runs of equal cost
        # (sharding.split_counts - the partition bench.py's steps use), so a rank owns whole contigs plus at most two
        # partial ones; a partial one is read through the index as a REGION of the contig (source.region_contig).
        if world > 1:
            units = sharding.split_counts({c: len(plan[c][0]) for c in names}, world)
        else:
            owner = sharding.lpt_assign({c: float(len(plan[c][0])) for c in names}, world)
            units = [(owner[c], c, 0, len(plan[c][0])) for c in names]
        mine = [(c, i0, i1) for r, c, i0, i1 in units if r == rank]
        whole = [c for c, i0, i1 in mine if i0 == 0 and i1 == len(plan[c][0])]
        local = {}
        # the bins' G + C: counted on a second context by a worker thread, in the order the units are taken below
        # (whole contigs in file order ~ chrom.sizes order, then the partial ones)
        gc_jobs = []
        for contig, i0, i1 in sorted(mine, key=lambda u: not (u[1] == 0 and u[2] == len(plan[u[0]][0]))):
            starts, stops, _arms, live, ok = plan[contig]
            idx, okp = _unit_bins(live, ok, i0, i1)
            if len(idx) and okp.any():
                gc_jobs.append(((contig, i0, i1), contig, starts[idx][okp], stops[idx][okp]))
        gc_ahead = _GCAhead(ref, eng, gc_jobs)
        # contigs are counted as they become resident: a file without a usable index is decoded in ONE streaming
        # pass (decode of contig k+1 beside the kernels / the reference upload of contig k); an indexed file is
        # read contig by contig through the index, so a rank touches only its own blocks
        err = None
        try:
            tw = time.perf_counter()
            for src, contig in (early if early is not None else resident_contigs(input_file, whole, workers,
                                                                                 stream_all=world == 1)):
                clock["decode_wait"] += time.perf_counter() - tw
                if contig not in plan:  # (the early stream hands out every contig chrom.sizes lists)
                    tw = time.perf_counter()
                    continue
                starts, stops, arms, live, ok = plan[contig]
                local[(contig, 0, len(starts))] = _contig_counts(src, eng, ref, contig, starts, stops, live, ok,
                                                                 contig_gaps.get(contig) if gaps is not None else None,
                                                                 blacklist, quality_threshold, clock, gc_ahead=gc_ahead)
                tw = time.perf_counter()
            clock["decode_wait"] += time.perf_counter() - tw
            for contig, i0, i1 in mine:
                starts, stops, arms, live, ok = plan[contig]
                if (contig, i0, i1) in local:
                    continue
                if i0 == 0 and i1 == len(starts):
                    # a planned contig the file does not hold: pysam raises for the unknown region
                    if live.any():
                        raise ValueError(f"could not create iterator for region '{contig}': contig not present in "
                                         f"{input_file}")
                    local[(contig, i0, i1)] = np.zeros((0, 4), np.int64)
                    continue
                if not live[i0:i1].any():
                    local[(contig, i0, i1)] = np.zeros((0, 4), np.int64)
                    continue
                tw = time.perf_counter()
                sel = np.nonzero(live[i0:i1])[0] + i0
                src, key = region_contig(input_file, contig, int(starts[sel].min()), int(stops[sel].max()), workers)
                clock["decode_wait"] += time.perf_counter() - tw
                try:
                    local[(contig, i0, i1)] = _contig_counts(src, eng, ref, contig, starts, stops, live, ok,
                                                             contig_gaps.get(contig) if gaps is not None else None,
                                                             blacklist, quality_threshold, clock, key=key, i0=i0, i1=i1,
                                                             gc_ahead=gc_ahead)
                finally:
                    if hasattr(src, "release_region"):
                        src.release_region(key)
        except Exception as e:  # noqa: BLE001 - with several ranks every rank must learn of it (sharding.agree)
            err = e
        finally:
            gc_ahead.close(abort=err is not None)  # (the worker reads the reference file: joined before it is closed)
        if world > 1:
            sharding.agree(err)
        elif err is not None:
            raise err
    tg = time.perf_counter()
    n_rows = {(c, i0, i1): int(plan[c][3][i0:i1].sum()) for _, c, i0, i1 in units}
    counts = sharding.gather_unit_rows(local, units, n_rows, 4)
    for c in names:
        counts.setdefault(c, np.zeros((0, 4), np.int64))
    clock["gather"] = time.perf_counter() - tg
    tg = time.perf_counter()
    # a contig listed twice in chrom.sizes has its bins once per listing (the reference loops over the listing,
    # frag/_delfi.py:269-283); it is counted once and laid down as often as it is listed
    listing = [c for c, _size in contigs if c in plan]
    parts = [_contig_columns(c, plan[c][0], plan[c][1], plan[c][2], plan[c][3], plan[c][4], counts[c]) for c in listing]
    window_df = _window_frame(parts)
    trimmed = window_df.loc[window_df["arm"] != "NOARM", :].copy()
    trimmed["ratio"] = np.where(trimmed["long"] == 0, np.nan, trimmed["short"] / trimmed["long"])

    if remove_nocov:  # the two hg19 no-coverage windows, by position (:333-340)
        pos = np.arange(trimmed.shape[0])
        final = trimmed.loc[np.logical_and(pos != 8779, pos != 13664)].reset_index()
    else:
        final = trimmed
    clock["frame"] = time.perf_counter() - tg
    if gc_correct:
        final = delfi_gc_correct(final, 0.75, 8, verbose)
    if merge_bins:
        tg = time.perf_counter()
        final = delfi_merge_bins(final, gc_correct, verbose=verbose)
        clock["merge"] = time.perf_counter() - tg
    if output_file is not None and rank == 0:  # every rank holds the same frame; one of them writes it
        tg = time.perf_counter()
        _write_delfi(final, output_file)
        clock["write"] = time.perf_counter() - tg
    clock["total"] = time.perf_counter() - t_begin
    LAST_STAGE_S.clear()
    LAST_STAGE_S.update({k: round(v, 4) for k, v in clock.items()})
    if verbose:
        stderr.write(f"{int(window_df['num_frags'].sum()) if len(window_df) else 0} fragments included.\n")
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
