"""
TEST INFRASTRUCTURE -- runs ONLY in the build container (needs /root/reference).

Golden vectors for ``agg_bw``: imports the reference's ``utils/_agg_bw.py`` through oracle/refstub.py with
an in-memory pyBigWig stand-in (pyBigWig is not installed here) and records what the reference returns and
writes for a seeded track and a strand-annotated interval file.

The stand-in implements what the reference relies on (utils/_agg_bw.py:80-101): ``values(chrom, start,
stop)`` returns one float per base, NaN where the track has no entry, and raises RuntimeError for an
unknown contig or bounds outside it.  The same track is written as a real bigWig with the product's writer
(values are float32 in a bigWig, so the stand-in rounds to float32 as well).

    tests/golden/aggbw_track.bw      the track
    tests/golden/aggbw_sites.bed     BED6 intervals (+ / - / . strands, wrong sizes, out of bounds, unknown contig)
    tests/golden/aggbw.npz / .json   per-case returned arrays, WIG text and parameters

Usage:  python oracle/gen_golden_aggbw.py
"""
from __future__ import annotations

import io
import json
import os
import sys
from contextlib import redirect_stdout

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import refstub  # noqa: E402

refstub.install()

GOLD = os.path.join(ROOT, "tests", "golden")
CHROMS = {"chrA": 60_000, "chrB": 20_000}
TRACK = {}  # contig -> list of (start, float32 values)


class _BW:
    def __init__(self, path, mode="r"):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def values(self, chrom, start, stop):
        if chrom not in CHROMS or start < 0 or stop > CHROMS[chrom] or start >= stop:
            raise RuntimeError("Invalid interval bounds!")
        out = np.full(stop - start, np.nan)
        for s0, vals in TRACK.get(chrom, []):
            lo, hi = max(start, s0), min(stop, s0 + len(vals))
            if hi > lo:
                out[lo - start:hi - start] = vals[lo - s0:hi - s0]
        return out.tolist()


sys.modules["pyBigWig"].open = lambda path, mode="r": _BW(path, mode)

import finaletoolkit.utils._agg_bw as RA  # noqa: E402  (the reference)


def main():
    from finaletoolkit_amd.bigwig import write_fixed_step_bigwig
    rng = np.random.default_rng(123)
    sig = lambda n: (rng.integers(-40, 41, n) + 25 * np.sin(np.arange(n) / 17.0)).round().astype(np.float32)
    TRACK["chrA"] = [(1_000, sig(20_000)), (30_000, sig(12_000))]
    TRACK["chrB"] = [(0, sig(9_000))]
    write_fixed_step_bigwig(os.path.join(GOLD, "aggbw_track.bw"), list(CHROMS.items()),
                            [(c, s0, v.astype(np.float64)) for c in CHROMS for s0, v in TRACK[c]])
    W = 400
    sites = [("chrA", 2_000, 2_000 + W, "+"), ("chrA", 5_000, 5_000 + W, "-"), ("chrA", 20_900, 20_900 + W, "+"),  # runs off the track: NaN -> 0
             ("chrA", 31_000, 31_000 + W, "."), ("chrA", 33_000, 33_000 + W, "-"), ("chrA", 35_000, 35_000 + W + 10, "+"),  # wrong size
             ("chrA", 59_800, 59_800 + W, "+"),  # beyond the contig -> RuntimeError -> skipped
             ("chrB", 100, 100 + W, "-"), ("chrB", 8_900, 8_900 + W, "+"), ("chrZ", 0, W, "+")]
    with open(os.path.join(GOLD, "aggbw_sites.bed"), "w") as fh:
        for c, a, b, st in sites:
            fh.write(f"{c}\t{a}\t{b}\tsite\t0\t{st}\n")
    cases = [dict(key="default"), dict(key="w120", median_window_size=120), dict(key="w121_mean", median_window_size=121, mean=True),
             dict(key="w0", median_window_size=0), dict(key="w2_mean", median_window_size=2, mean=True)]
    A, meta = {}, []
    for cs in cases:
        kw = {k: v for k, v in cs.items() if k != "key"}
        out = os.path.join(GOLD, "_aggbw_tmp.wig")
        with redirect_stdout(io.StringIO()) as printed, np.errstate(all="ignore"):
            got = RA.agg_bw(os.path.join(GOLD, "aggbw_track.bw"), os.path.join(GOLD, "aggbw_sites.bed"), out, **kw)
        A[cs["key"]] = np.asarray(got)
        meta.append(dict(key=cs["key"], kwargs=kw, dtype=str(np.asarray(got).dtype), wig=open(out).read(),
                         printed_lines=len(printed.getvalue().splitlines())))
        os.unlink(out)
    np.savez_compressed(os.path.join(GOLD, "aggbw.npz"), **A)
    with open(os.path.join(GOLD, "aggbw.json"), "w") as fh:
        json.dump(meta, fh, indent=1)
    print("wrote", len(cases), "cases;", {k: (v.dtype, v.shape) for k, v in A.items()})


if __name__ == "__main__":
    main()
