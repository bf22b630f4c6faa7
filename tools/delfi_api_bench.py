#!/usr/bin/env python3
"""API-level DELFI (GPU box): frag.delfi() on a multi-contig 30x fragment file with 100 kb bins, blacklist, gap
annotation and a 2bit reference (GC per bin on the device), merged to 5 Mb -- wall time per call and a cProfile of the
host side.  usage: tools/delfi_api_bench.py [contigs=19,20,21,22]"""
import cProfile
import io
import json
import os
import pstats
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from finaletoolkit_amd import bgzf, frag, source, synth  # noqa: E402
from tests import helpers as H  # noqa: E402

names = (sys.argv[1] if len(sys.argv) > 1 else "19,20,21,22").split(",")
tmp = tempfile.mkdtemp(prefix="ftk_delfi_")
dev = torch.device("cuda", 0)
rows = []
for k, c in enumerate(names):
    size = synth.B37_SIZES[c]
    s, e, q, st = (t.cpu().numpy() for t in synth.gen_contig_device(torch, dev, size, synth.n_fragments(size, 30.0), 900 + k))
    rows.append((c, s, e, q, st))
path = os.path.join(tmp, "g.frag.gz")
bgzf.write_frag_gz(path, rows, level=1, with_index=True)
del rows
open(os.path.join(tmp, "cs.genome"), "w").write("".join(f"{c}\t{synth.B37_SIZES[c]}\n" for c in names))
open(os.path.join(tmp, "bins.bed"), "w").write("".join(
    f"{c}\t{a}\t{min(a + 99_999, synth.B37_SIZES[c])}\n" for c in names for a in range(0, synth.B37_SIZES[c], 100_000)))
rng = np.random.default_rng(3)
seqs = {c: np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, synth.B37_SIZES[c])].tobytes().decode() for c in names}
H.write_2bit(os.path.join(tmp, "ref.2bit"), seqs)
del seqs
gaps = []
for c in names:
    n = synth.B37_SIZES[c]
    c0 = int(n * 0.4) // 100_000 * 100_000
    gaps += [f"{c}\t0\t10000\ttelomere\n", f"{c}\t{c0}\t{c0 + 3_000_000}\tcentromere\n", f"{c}\t{n - 10000}\t{n}\ttelomere\n"]
open(os.path.join(tmp, "gaps.bed"), "w").write("".join(gaps))
bl = []
for k, c in enumerate(names):
    bs, be = synth.synth_blacklist(synth.B37_SIZES[c], 5 + k, 200)
    bl += [f"{c}\t{a}\t{b}\n" for a, b in zip(bs.tolist(), be.tolist())]
open(os.path.join(tmp, "bl.bed"), "w").write("".join(bl))


def call():
    return frag.delfi(path, os.path.join(tmp, "cs.genome"), os.path.join(tmp, "bins.bed"), os.path.join(tmp, "ref.2bit"),
                      blacklist_file=os.path.join(tmp, "bl.bed"), gap_file=os.path.join(tmp, "gaps.bed"), no_gc_correct=True,
                      remove_nocov=False, merge_bins=True, output_file=os.path.join(tmp, "delfi.tsv"))


res = {"contigs": names, "bins": sum(-(-synth.B37_SIZES[c] // 100_000) for c in names), "calls_s": []}
for rep in range(3):
    source.close_all()
    t0 = time.perf_counter()
    df = call()
    res["calls_s"].append(round(time.perf_counter() - t0, 3))
res["merged_rows"] = int(len(df))
source.close_all()
pr = cProfile.Profile()
pr.enable()
call()
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumtime").print_stats(18)
res["profile"] = out.getvalue().splitlines()[:40]
print(json.dumps(res, indent=1))
