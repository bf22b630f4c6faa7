"""
TEST INFRASTRUCTURE -- runs ONLY in the build container (needs /root/reference).

Golden vectors for ``adjust_wps``: imports the reference's
``frag/_adjust_wps.py`` through oracle/refstub.py with an in-memory pyBigWig
stand-in (pyBigWig is not installed here) and records what the reference hands
to ``addEntries`` for a seeded synthetic raw-WPS track.

The pyBigWig stand-in implements the behaviour the reference relies on
(frag/_adjust_wps.py:79-101, 300-318): ``intervals(chrom, start, end)`` returns
the entries overlapping the range as ``(start, end, float32 value)`` tuples,
``None`` when there are none, and raises RuntimeError for an unknown contig or
bounds outside it; a file opened for writing records ``addHeader`` /
``addEntries``.

    tests/golden/adjust_wps.npz    the raw track + per-case outputs
    tests/golden/adjust_sites.bed  interval file
    tests/golden/adjust_wps.json   case parameters

Usage:  python oracle/gen_golden_adjust.py
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)

import refstub  # noqa: E402

refstub.install()

GOLD = os.path.join(ROOT, "tests", "golden")
CHROMS = {"chrA": 400_000, "chrB": 150_000}

# the raw track: runs of consecutive positions, integer-valued like WPS
TRACK = {}        # contig -> list of (start, values float64)
WRITTEN = []      # (contig, starts, ends, values) per addEntries call


class _BW:
    def __init__(self, path, mode="r"):
        self.mode = mode

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def close(self):
        pass

    def intervals(self, chrom, start, stop):
        if chrom not in CHROMS or start < 0 or stop > CHROMS[chrom] or start >= stop:
            raise RuntimeError("Invalid interval bounds!")
        out = []
        for s0, vals in TRACK.get(chrom, []):
            lo, hi = max(start, s0), min(stop, s0 + len(vals))
            for p in range(lo, hi):
                out.append((p, p + 1, float(np.float32(vals[p - s0]))))
        return tuple(out) if out else None

    def addHeader(self, header):
        self.header = header

    def addEntries(self, chroms, starts, ends=None, values=None):
        WRITTEN.append((chroms[0], np.asarray(starts, np.int64), np.asarray(ends, np.int64),
                        np.asarray(values, np.float64)))


sys.modules["pyBigWig"].open = lambda path, mode="r": _BW(path, mode)

import finaletoolkit.frag._adjust_wps as RA  # noqa: E402  (the reference)


class _InlinePool:
    def __init__(self, n):
        pass

    def imap(self, fn, it):
        return map(fn, it)

    def close(self):
        pass


RA.Pool = _InlinePool  # same results as multiprocessing.Pool.imap, in order


def wps_like(rng, n):
    """Integer-valued, autocorrelated, WPS-like signal."""
    steps = rng.integers(-3, 4, n)
    v = np.cumsum(steps)
    v = v - np.round(np.convolve(v, np.ones(301) / 301, "same"))
    return (v + 20 * np.sin(np.arange(n) / 30.0)).round().astype(np.float64)


def main():
    rng = np.random.default_rng(77)
    TRACK["chrA"] = [(0, wps_like(rng, 9_000)), (40_000, wps_like(rng, 22_000)), (100_000, wps_like(rng, 3_000))]
    TRACK["chrB"] = [(60_000, wps_like(rng, 5_000)), (149_000, wps_like(rng, 1_000))]
    A = {}
    for c, runs in TRACK.items():
        for k, (s0, v) in enumerate(runs):
            A[f"track_{c}_{k}_start"] = np.int64(s0)
            A[f"track_{c}_{k}_values"] = v
    sites = [("chrA", 1_000, 1_200),      # start clipped to 0
             ("chrA", 42_400, 42_600), ("chrA", 45_000, 45_100),   # these two merge at interval_size 5000 / W 1000
             ("chrA", 55_000, 55_010),
             ("chrA", 101_400, 101_600),  # 3000-wide run: fits interval_size 3000 only
             ("chrA", 200_000, 200_100),  # no entries -> skipped
             ("chrB", 62_500, 62_500),
             ("chrB", 149_500, 149_600),  # stop beyond the contig -> RuntimeError -> skipped
             ("chrZ", 10, 20)]            # unknown contig -> skipped
    with open(os.path.join(GOLD, "adjust_sites.bed"), "w") as fh:
        for c, a, b in sites:
            fh.write(f"{c}\t{a}\t{b}\n")
    with open(os.path.join(GOLD, "adjust.chrom.sizes"), "w") as fh:
        for c, n in CHROMS.items():
            fh.write(f"{c}\t{n}\n")
    cases = [
        dict(key="default3k", interval_size=3000),
        dict(key="w200_nosavgol", interval_size=3000, median_window_size=200, savgol=False),
        dict(key="mean_edges", interval_size=3000, median_window_size=600, mean=True, subtract_edges=True,
             edge_size=300),
        dict(key="median_edges_sg", interval_size=3000, median_window_size=500, subtract_edges=True, edge_size=100,
             savgol_window_size=31, savgol_poly_deg=3),
        dict(key="merge5k", interval_size=5000, median_window_size=1000),
    ]
    for cs in cases:
        WRITTEN.clear()
        kw = {k: v for k, v in cs.items() if k != "key"}
        # the 3 kb run cannot hold a 5 kb interval: the reference raises there, so case merge5k uses a
        # site file without it
        bed = os.path.join(GOLD, "adjust_sites.bed")
        if cs["key"] == "merge5k":
            bed = os.path.join(GOLD, "adjust_sites_5k.bed")
            with open(bed, "w") as fh:
                for c, a, b in sites:
                    if (c, a) not in (("chrA", 101_400), ("chrB", 62_500)):
                        fh.write(f"{c}\t{a}\t{b}\n")
        RA.adjust_wps("raw.bw", bed, "out.bw", os.path.join(GOLD, "adjust.chrom.sizes"), **kw)
        cs["n_runs"] = len(WRITTEN)
        cs["run_contigs"] = [w[0] for w in WRITTEN]
        for i, (c, st, en, v) in enumerate(WRITTEN):
            assert np.all(en == st + 1) and np.all(np.diff(st) == 1)
            A[f"{cs['key']}_{i}_start"] = np.int64(st[0])
            A[f"{cs['key']}_{i}_values"] = v
    # non-integer data through the reference's filter helpers (exact median on arbitrary doubles)
    x = rng.normal(0, 5, 2_600)
    x[100:110] = x[100]          # ties
    x[500] = -0.0
    pos, adj = RA._median_filter(np.arange(2_600), x, 400)
    A["float_input"] = x
    A["float_median400"] = adj
    A["float_pos0"] = np.int64(pos[0])
    pos, adj = RA._mean_filter(np.arange(2_600), x, 400)
    A["float_mean400"] = adj
    with open(os.path.join(GOLD, "adjust_wps.json"), "w") as fh:
        json.dump(cases, fh, indent=0, sort_keys=True)
    np.savez_compressed(os.path.join(GOLD, "adjust_wps.npz"), **A)
    print("adjust_wps goldens written:", {c["key"]: c["n_runs"] for c in cases})


if __name__ == "__main__":
    main()
