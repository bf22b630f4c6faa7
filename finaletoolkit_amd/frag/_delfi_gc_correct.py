"""
LOESS GC correction of DELFI windows (reference: ``frag/_delfi_gc_correct.py:19-94``).

OUT OF THE MI355X HOT PATH: an O(n_bins) host fit done entirely by the
third-party ``loess`` package (``loess.loess_1d.loess_1d``, pinned 2.1.2 by the
reference).  That package is not part of this repository; when it is not
installed, requesting GC correction raises ImportError (use
``no_gc_correct=True``).  Parity of this step is unpinned.
"""
from __future__ import annotations

import numpy as np
from .._lazy import LazyModule

pandas = LazyModule("pandas")

__all__ = ["delfi_gc_correct"]

_COLUMNS = ["short", "long", "num_frags", "ratio"]


def delfi_gc_correct(windows: pandas.DataFrame, alpha: float = 0.75, it: int = 8, verbose: bool = False):
    try:
        from loess.loess_1d import loess_1d
    except ImportError as e:  # pragma: no cover - depends on the environment
        raise ImportError("DELFI GC correction needs the third-party 'loess' package (loess.loess_1d); "
                          "install it or call delfi(..., no_gc_correct=True)") from e
    out = windows.copy()
    out.replace([np.inf, -np.inf], np.nan, inplace=True)
    valid = out.dropna()
    gc_range = np.arange(valid["gc"].min(), valid["gc"].max() + 0.01, 0.01)
    for column in _COLUMNS:
        _, line, _ = loess_1d(valid["gc"].to_numpy(), valid[column].to_numpy(), xnew=gc_range, degree=2, frac=alpha)
        out[f"{column}_corrected"] = out[column] - np.interp(out["gc"], gc_range, line) + valid[column].median()
    return out
