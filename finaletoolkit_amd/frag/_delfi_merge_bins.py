"""
5 Mb DELFI windows from 100 kb bins, arm by arm (host pandas, O(n_bins)).

Behavioural contract = the reference's ``frag/_delfi_merge_bins.py:13-92``:
fifty consecutive bins of one arm make a window; a p-arm is cut from its first
bin (the incomplete tail is dropped), a q-arm from its LAST bin (the incomplete
head is dropped), so both arms are anchored at the telomere-far side the
original DELFI scripts used.  Counts are summed; ``gc`` and ``ratio`` are
per-window means.  Arms whose label holds neither ``p`` nor ``q`` are ignored.
"""
from __future__ import annotations

import numpy as np
import pandas as pd

__all__ = ["delfi_merge_bins"]

_N = 50  # bins per merged window
_TOTALS = ["short", "long", "num_frags"]
_TOTALS_GC = ["short_corrected", "long_corrected", "num_frags_corrected"]


def _arm_windows(arm_rows: pd.DataFrame, arm: str, skip: int, n_full: int, with_corrected: bool) -> dict:
    """All merged windows of one arm at once: every column as a ``[n_full, 50]`` block reduced along its rows.
    Same arithmetic as pandas on each fifty-row chunk: ``sum`` = numpy's sum of the fifty values, ``mean`` =
    NaN-skipping (``nanops.nanmean``: NaN replaced by 0, summed, divided by the count of the others; no value at
    all -> NaN), ``min`` / ``max`` of the coordinates."""
    def block(col):
        return np.ascontiguousarray(arm_rows[col].to_numpy()[skip: skip + n_full * _N]).reshape(n_full, _N)

    def nan_mean(col):
        v = block(col).astype(np.float64)
        mask = np.isnan(v)
        count = _N - mask.sum(axis=1)
        total = np.where(mask, 0.0, v).sum(axis=1)
        with np.errstate(invalid="ignore", divide="ignore"):
            return np.where(count > 0, total / np.maximum(count, 1), np.nan)

    def nan_sum(col):  # pandas sums skip NaN too (an all-NaN chunk sums to 0)
        v = block(col)
        if v.dtype.kind != "f":
            return v.sum(axis=1)
        return np.where(np.isnan(v), 0.0, v).sum(axis=1)

    names = np.empty(n_full, dtype=object)
    names[:] = arm[:-1]
    arms = np.empty(n_full, dtype=object)
    arms[:] = arm
    out = {"contig": names, "start": block("start").min(axis=1), "stop": block("stop").max(axis=1), "arm": arms,
           "short": nan_sum("short"), "long": nan_sum("long"), "gc": nan_mean("gc"), "num_frags": nan_sum("num_frags"),
           "ratio": nan_mean("ratio")}
    if with_corrected:
        for c in _TOTALS_GC:
            out[c] = nan_sum(c)
        out["ratio_corrected"] = nan_mean("ratio_corrected")
    return out


def delfi_merge_bins(hundred_kb_bins: pd.DataFrame, gc_corrected: bool = True, verbose: bool = False) -> pd.DataFrame:
    """Merged frame with the input's columns (minus a stray ``index`` column)."""
    columns = [c for c in hundred_kb_bins.columns if c != "index"]
    parts: list[dict] = []
    arm_col = hundred_kb_bins["arm"].to_numpy()
    for arm in pd.unique(hundred_kb_bins["arm"]):
        if "p" in arm:
            from_tail = False
        elif "q" in arm:
            from_tail = True
        else:
            continue
        arm_rows = hundred_kb_bins.loc[arm_col == arm]
        n_full = arm_rows.shape[0] // _N
        skip = arm_rows.shape[0] - n_full * _N if from_tail else 0
        if n_full:
            parts.append(_arm_windows(arm_rows, arm, skip, n_full, gc_corrected))
    if not parts:
        return pd.DataFrame([], columns=columns)
    return pd.DataFrame({c: np.concatenate([p[c] for p in parts]) for c in columns}, columns=columns)
