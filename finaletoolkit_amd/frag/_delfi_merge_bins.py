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

import pandas as pd

__all__ = ["delfi_merge_bins"]

_N = 50  # bins per merged window
_TOTALS = ["short", "long", "num_frags"]
_TOTALS_GC = ["short_corrected", "long_corrected", "num_frags_corrected"]


def _window_row(rows: pd.DataFrame, arm: str, with_corrected: bool) -> tuple:
    out = [arm[:-1], rows["start"].min(), rows["stop"].max(), arm, rows["short"].sum(), rows["long"].sum(),
           rows["gc"].mean(), rows["num_frags"].sum(), rows["ratio"].mean()]
    if with_corrected:
        out.extend(rows[c].sum() for c in _TOTALS_GC)
        out.append(rows["ratio_corrected"].mean())
    return tuple(out)


def delfi_merge_bins(hundred_kb_bins: pd.DataFrame, gc_corrected: bool = True, verbose: bool = False) -> pd.DataFrame:
    """Merged frame with the input's columns (minus a stray ``index`` column)."""
    columns = [c for c in hundred_kb_bins.columns if c != "index"]
    records: list[tuple] = []
    for arm in pd.unique(hundred_kb_bins["arm"]):
        if "p" in arm:
            from_tail = False
        elif "q" in arm:
            from_tail = True
        else:
            continue
        arm_rows = hundred_kb_bins.loc[hundred_kb_bins["arm"] == arm]
        n_full = arm_rows.shape[0] // _N
        skip = arm_rows.shape[0] - n_full * _N if from_tail else 0
        for k in range(n_full):
            records.append(_window_row(arm_rows.iloc[skip + k * _N: skip + (k + 1) * _N], arm, gc_corrected))
    return pd.DataFrame(records, columns=columns)
