#!/usr/bin/env python3
"""API-level throughput of the per-base features WITH an output file (GPU box): chr22 at 30x from a frag.gz
on disk through frag.wps(..., output_file=.wig / .wig.gz) and frag.multi_wps(10 261 x 5 kb sites -> .bw /
.bed.gz).  "python_statement_s" is the reference's writer statement (one f-string per base / pyBigWig-like
zlib per section in Python) timed on a 1 M-value sample of the same scores and scaled to the full output.
usage: tools/writer_bench.py > gpurun_out/writer_bench.json"""
import gzip
import json
import os
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from finaletoolkit_amd import bgzf, frag, source, synth  # noqa: E402

size = synth.B37_SIZES["22"]
tmp = tempfile.mkdtemp(prefix="ftk_wb_")
s, e, q, st = synth.synth_contig(size, 30.0, synth.SEED_BASE + 21)
path = os.path.join(tmp, "chr22.frag.gz")
bgzf.write_frag_gz(path, [("22", s, e, q, st)], level=1, with_index=True)
cs = os.path.join(tmp, "cs.genome")
open(cs, "w").write(f"22\t{size}\n")
sites = os.path.join(tmp, "sites.bed")
open(sites, "w").write("".join(f"22\t{a + 2400}\t{a + 2600}\n" for a in range(0, size - 5000, 5000)))
res = {"contig": "22", "fragments": len(s), "bases": size}


def timed(fn, reps=3):
    """Wall time of the call itself: the previous repetition's result (4 GB of 80-byte records for chr22, 0.15 s to
    unmap) is dropped before the clock starts - round 2's numbers had that inside the next call."""
    ts = []
    out = None
    for _ in range(reps):
        out = None
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return out, ts


scores, ts = timed(lambda: frag.wps(path, "22", 0, size, size))
res["wps_no_output_s"] = [round(t, 3) for t in ts]
for suffix in (".wig", ".wig.gz"):
    out = os.path.join(tmp, "chr22" + suffix)
    _, ts = timed(lambda: frag.wps(path, "22", 0, size, size, output_file=out))
    _ = None
    res["wps" + suffix + "_s"] = [round(t, 3) for t in ts]
    res["wps" + suffix + "_MB"] = round(os.path.getsize(out) / 1e6, 1)
# the file holds what the reference's writer would have written
with gzip.open(os.path.join(tmp, "chr22.wig.gz"), "rt") as fh:
    head = fh.readline()
    body = np.loadtxt(fh, dtype=np.int64, max_rows=200_000)
assert head == f"fixedStep\tchrom=22\tstart=0\tstep=1\tspan={size}\n" and np.array_equal(body, scores["wps"][:200_000])
sample = scores["wps"][20_000_000:21_000_000]
t0 = time.perf_counter()
"".join(f"{v}\n" for v in sample)
res["wig_python_statement_s"] = round((time.perf_counter() - t0) * size / len(sample), 1)
for suffix in (".bw", ".bed.gz"):
    out = os.path.join(tmp, "sites" + suffix)
    _, ts = timed(lambda: frag.multi_wps(path, sites, chrom_sizes=cs, output_file=out))
    res["multi_wps" + suffix + "_s"] = [round(t, 3) for t in ts]
    res["multi_wps" + suffix + "_MB"] = round(os.path.getsize(out) / 1e6, 1)
n_sites = sum(1 for _ in open(sites))
res["sites"] = n_sites
t0 = time.perf_counter()
pos = np.arange(len(sample))
with gzip.open(os.path.join(tmp, "py.bed.gz"), "wt") as fh:
    fh.write("".join(f"22\t{p}\t{p + 1}\t{v}\n" for p, v in zip(pos.tolist(), sample.tolist())))
res["bedgraph_gz_python_statement_s"] = round((time.perf_counter() - t0) * n_sites * 5000 / len(sample), 1)
t0 = time.perf_counter()
for k in range(0, len(sample), 5000):
    zlib.compress(sample[k:k + 5000].astype(np.float64).astype("<f4").tobytes(), 6)
res["bigwig_python_sections_s"] = round((time.perf_counter() - t0) * n_sites * 5000 / len(sample), 1)
# and the bigWig reads back as the scores
from finaletoolkit_amd.bigwig import BigWigFile  # noqa: E402
bw = BigWigFile(os.path.join(tmp, "sites.bw"))
assert np.array_equal(bw.values("22", 10_000, 15_000), scores["wps"][10_000:15_000].astype(np.float64))
source.close_all()
print(json.dumps(res))
