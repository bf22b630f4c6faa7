"""
TEST INFRASTRUCTURE ONLY -- the checker of BASELINE config 5 at real size (a > 4 GiB, multi-contig 60x BAM streamed
through the device parser) and of its fragment-file twin (a > 4 GiB whole-genome frag.gz).  Shared by
``tests/test_gpu_bam_scale.py``, ``tests/test_gpu_text_scale.py`` and the untimed check of ``bench.py``'s
``bam_60x_chr1_scale`` leg; never the thing measured, never imported by the product.

Reference semantics being checked (``io/alignment.py:242-268`` behind ``utils/_frag_generator.py:58-141``): a BAM
window query returns the read1 alignments overlapping the window, so every count here is taken by the C oracle in
read1-fetch mode (``Frags`` with the read1 span).
"""
from __future__ import annotations

import numpy as np

from . import oracle as O

WINDOW = 100_000


def sample_windows(n_win: int, want: int) -> np.ndarray:
    """``want`` window indices spread over the contig, always with the first and the last two (the partial bin)."""
    if n_win <= want:
        return np.arange(n_win)
    step = max(1, n_win // (want - 3))
    return np.unique(np.concatenate([np.arange(0, n_win, step), [n_win - 2, n_win - 1]]))


def wps_ranges(size: int, length: int = 50_000):
    """Three ranges per contig: its first bases, the middle, and the last ones."""
    length = min(length, size)
    return [(0, length), (max(0, size // 2 - length // 2), max(0, size // 2 - length // 2) + length), (size - length, size)]


def frags_of(exp) -> O.Frags:
    """(a fragment FILE's contig has no read1 span: tabix-overlap semantics, ``io/alignment.py:270-302``)"""
    return O.Frags(exp["s"], exp["e"], exp["q"], exp["st"], exp.get("r1s"), exp.get("r1e"))


def check_contig(eng, key, size, exp, features, n_sampled=24, wps_len=50_000):
    """One resident contig against the oracle: exact fragment count; ``n_sampled`` of its 100 kb windows (coverage,
    1001-bin length histogram + overflow, DELFI short / long / fragments) out of ``features`` = the engine's
    ``window_features`` over ALL tiling windows; three ``wps_len`` ranges of per-base WPS, engine and oracle on the
    same interval.  Returns ``(ok, detail)``."""
    from finaletoolkit_amd import synth
    fr = frags_of(exp)
    ws, we = synth.tiling_windows(size, WINDOW)
    pick = sample_windows(len(ws), n_sampled)
    detail = {"fragments": int(eng.info(key)[0]), "fragments_expected": int(exp["n"]), "windows_checked": int(len(pick))}
    ok = detail["fragments"] == detail["fragments_expected"]
    cov = O.c_window_counts(fr, ws[pick], we[pick], mapq_min=30)
    h, ov = O.c_fraglen_hist(fr, ws[pick], we[pick], 0, 1001, mapq_min=30)
    sh, lg, nf = O.c_delfi_counts(fr, ws[pick], we[pick], 30)
    detail["coverage_ok"] = bool(np.array_equal(features["coverage"][pick], cov) and int(cov.sum()) > 0)
    detail["histogram_ok"] = bool(np.array_equal(features["hist"][pick], h) and np.array_equal(features["overflow"][pick], ov))
    detail["delfi_ok"] = bool(np.array_equal(features["short"][pick], sh) and np.array_equal(features["long"][pick], lg)
                              and int(sh.sum() + lg.sum()) == int(nf.sum()))
    bases = 0
    wps_ok = True
    for a, b in wps_ranges(size, wps_len):
        got = eng.wps(key, a, b, size, 120, 120, 180, 30)
        wps_ok = wps_ok and bool(np.array_equal(got, O.c_wps(fr, a, b, size, 120, 120, 180, 30)))
        bases += b - a
    detail["wps_ok"], detail["wps_bases_checked"] = wps_ok, bases
    ok = ok and detail["coverage_ok"] and detail["histogram_ok"] and detail["delfi_ok"] and wps_ok
    return bool(ok), detail


def wps_closed_form_sum(exp, size) -> int:
    """Sum of the whole contig's WPS (W=120, lengths 120..180, mapq >= 30) when every read1 overlaps the contig-wide
    fetch window: a passing fragment gives 1 to [fs+61, fe-60] and takes 1 from [fs-59, fs+60] and [fe-59, fe+60], each
    clipped to the contig (derived closed forms)."""
    s, e, q = exp["s"], exp["e"], exp["q"]
    keep = (q >= 30) & (e - s >= 120) & (e - s <= 180)
    fs, fe = s[keep].astype(np.int64), e[keep].astype(np.int64)

    def clipped(a, b):
        return np.maximum(np.minimum(b, size - 1) - np.maximum(a, 0) + 1, 0)
    return int((clipped(fs + 61, fe - 60) - clipped(fs - 59, fs + 60) - clipped(fe - 59, fe + 60)).sum())


def check_region(eng, key, size, exp, start, stop, pad=400):
    """A REGION table (``ftk_fragstream_open_region``: every fragment whose read1 overlaps ``[start, stop)``) against the
    oracle run on the WHOLE contig's fragments: the 100 kb windows inside the region, and the WPS of its inner bases
    (``pad`` in from both ends: an interval call's fetch window reaches max_length beyond it)."""
    fr = frags_of(exp)
    lo = -(-start // WINDOW) * WINDOW
    ws = np.arange(lo, stop - WINDOW + 1, WINDOW, dtype=np.int32)
    we = (ws + WINDOW).astype(np.int32)
    f = eng.window_features(key, ws, we, 30, hist=(0, 1001), delfi=dict(quality_threshold=30))
    h, ov = O.c_fraglen_hist(fr, ws, we, 0, 1001, mapq_min=30)
    sh, lg, _ = O.c_delfi_counts(fr, ws, we, 30)
    ok = len(ws) > 0 and np.array_equal(f["coverage"], O.c_window_counts(fr, ws, we, mapq_min=30))
    ok = ok and np.array_equal(f["hist"], h) and np.array_equal(f["overflow"], ov)
    ok = ok and np.array_equal(f["short"], sh) and np.array_equal(f["long"], lg) and int(f["coverage"].sum()) > 0
    a, b = start + pad, min(stop - pad, start + pad + 60_000)
    ok = ok and np.array_equal(eng.wps(key, a, b, size, 120, 120, 180, 30), O.c_wps(fr, a, b, size, 120, 120, 180, 30))
    n_region = int(eng.info(key)[0])
    return bool(ok), {"region_rows": n_region, "contig_rows": int(exp["n"]), "windows_checked": int(len(ws)),
                      "wps_bases_checked": int(b - a)}


def region_file_offset(exp, start) -> int:
    """File offset (compressed bytes) the ``.bai`` linear index sends a reader to for position ``start``."""
    return int(exp["linear"][start >> 14] >> np.uint64(16))
