"""Shared test helpers (independent of the product decoder)."""
import gzip
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def read_frag_gz(path):
    """Parse a FinaleDB / BED6 fragment file with Python's gzip: {contig: (start, end, mapq, strand)}."""
    rows = {}
    bed6 = None
    with gzip.open(path, "rt") as fh:
        for line in fh:
            if not line.strip() or line.startswith("#"):
                continue
            p = line.rstrip("\n").split("\t")
            if bed6 is None:
                bed6 = len(p) > 5
            mq, st = (p[4], p[5]) if bed6 else (p[3], p[4])
            rows.setdefault(p[0], []).append((int(p[1]), int(p[2]), int(mq), 1 if "+" in st else 0))
    out = {}
    for c, r in rows.items():
        a = np.array(r, dtype=np.int64)
        out[c] = (a[:, 0].astype(np.int32), a[:, 1].astype(np.int32), a[:, 2].astype(np.uint8),
                  a[:, 3].astype(np.uint8))
    return out


def golden_json():
    with open(os.path.join(GOLDEN, "golden.json")) as fh:
        return json.load(fh)


def golden_npz():
    return np.load(os.path.join(GOLDEN, "golden.npz"))


def read_bed(path):
    """BED reader with the reference's skipping rules (utils/utils.py:310-343)."""
    out = []
    with open(path) as fh:
        for line in fh:
            if line.startswith(("#", "track", "browser")) or not line.strip():
                continue
            p = line.strip().split("\t")
            if len(p) < 3:
                continue
            out.append((p[0], int(p[1]), int(p[2]), p[3] if len(p) > 3 else "."))
    return out
