#!/usr/bin/env python3
"""Where a small file's 20 ms go: chr22 at 30x (33 MB) through source.stream_source + features + WPS, three timed
repetitions with marks (resident / features / wps), then one under cProfile (the Python side: stream_source,
load_contig_from_table - a 50 MB hipMalloc is 3 ms -, wps)."""
import os, sys, time, cProfile, pstats, tempfile
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import numpy as np
from finaletoolkit_amd import source, synth, bgzf
import bench
dev = torch.device("cuda", 0)
size = synth.B37_SIZES["22"]
n = synth.n_fragments(size, 30.0)
s, e, q, st = (t.cpu().numpy() for t in bench.gen_contig_device(torch, dev, size, n, 5))
tmp = tempfile.mkdtemp()
p = os.path.join(tmp, "c22.frag.gz")
bgzf.write_frag_gz(p, [("22", s, e, q, st)], level=1, with_index=False)
ws, we = synth.tiling_windows(size, 100_000)
threads = source.usable_cores()
def run():
    t0 = time.perf_counter()
    marks = []
    for src, c in source.stream_source(p, threads):
        marks.append(("resident", time.perf_counter() - t0))
        eng = source.get_engine()
        r = eng.window_features(src.key(c), ws, we, 30, hist=(0, 1001), delfi=dict(quality_threshold=30))
        marks.append(("features", time.perf_counter() - t0))
        w = eng.wps(src.key(c), 0, size, size)
        marks.append(("wps", time.perf_counter() - t0))
    return marks
for _ in range(3):
    source.close_all(); source.get_engine()
    print([(k, round(v * 1e3, 2)) for k, v in run()])
source.close_all(); source.get_engine()
pr = cProfile.Profile(); pr.enable(); run(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
