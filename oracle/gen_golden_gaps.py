"""
TEST INFRASTRUCTURE -- runs ONLY in the build container (needs /root/reference).

sha256 of the BED4 files the reference's ``gap-bed`` command writes for its three bundled gap tracks
(genome/gaps.py:270-302), imported through oracle/refstub.py -> tests/golden/gap_bed_sha256.json.

Usage:  python oracle/gen_golden_gaps.py
"""
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)

import refstub  # noqa: E402

refstub.install()
from finaletoolkit.genome.gaps import _cli_gap_bed  # noqa: E402  (the reference)

out = {}
tmp = tempfile.mkdtemp()
for genome in ("hg19", "b37", "human_g1k_v37", "hg38", "GRCh38"):
    path = os.path.join(tmp, genome + ".bed")
    _cli_gap_bed(genome, path)
    data = open(path, "rb").read()
    out[genome] = {"sha256": hashlib.sha256(data).hexdigest(), "bytes": len(data), "lines": data.count(b"\n")}
with open(os.path.join(ROOT, "tests", "golden", "gap_bed_sha256.json"), "w") as fh:
    json.dump(out, fh, indent=1)
print(out)
