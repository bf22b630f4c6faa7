"""
Merge 100 kb DELFI bins into 5 Mb (50-bin) windows per chromosome arm -- the
semantics of the reference's ``frag/_delfi_merge_bins.py:13-92``: p-arms are
chunked 5'->3' from the first bin; q-arms from the LAST bin backwards (the loop
stops before index 0) and then reversed; incomplete chunks are dropped; counts
are summed, ``gc`` / ``ratio`` averaged.  Host pandas code: O(n_bins).
"""
from __future__ import annotations

import pandas as pd

__all__ = ["delfi_merge_bins"]

_BINS_PER_WINDOW = 50
_SUM = ("short", "long", "num_frags")
_SUM_CORRECTED = ("short_corrected", "long_corrected", "num_frags_corrected")


def _merge(chunk: pd.DataFrame, arm: str, gc_corrected: bool) -> tuple:
    rec = [arm[:-1], chunk["start"].min(), chunk["stop"].max(), arm]
    rec += [chunk["short"].sum(), chunk["long"].sum(), chunk["gc"].mean(), chunk["num_frags"].sum(),
            chunk["ratio"].mean()]
    if gc_corrected:
        rec += [chunk[c].sum() for c in _SUM_CORRECTED] + [chunk["ratio_corrected"].mean()]
    return tuple(rec)


def delfi_merge_bins(hundred_kb_bins: pd.DataFrame, gc_corrected: bool = True, verbose: bool = False) -> pd.DataFrame:
    merged: list[tuple] = []
    for arm in hundred_kb_bins["arm"].unique():
        bins = hundred_kb_bins[hundred_kb_bins["arm"] == arm].reset_index()
        n = bins.shape[0]
        if "p" in arm:
            for lo in range(0, n, _BINS_PER_WINDOW):
                chunk = bins.iloc[lo:lo + _BINS_PER_WINDOW]
                if chunk.shape[0] == _BINS_PER_WINDOW:
                    merged.append(_merge(chunk, arm, gc_corrected))
        elif "q" in arm:
            tail_first: list[tuple] = []
            for hi in range(n - 1, 0, -_BINS_PER_WINDOW):
                lo = hi - (_BINS_PER_WINDOW - 1)
                if lo < 0:
                    continue
                tail_first.append(_merge(bins.iloc[lo:hi + 1], arm, gc_corrected))
            merged.extend(reversed(tail_first))
    return pd.DataFrame(merged, columns=hundred_kb_bins.columns[hundred_kb_bins.columns != "index"])
