#!/usr/bin/env python3
"""BASELINE config 5 on one GPU, end to end: a 60x coordinate-sorted paired-end BAM slice on disk -> streaming
decoder (BGZF inflate, record chain, read1 fragments, sort by start: host threads) -> HBM -> coverage + 1001-bin
histogram + DELFI + per-base WPS -> results in host memory; stage times of the decoder's producer thread.
usage: tools/bam_e2e_bench.py [slice_bp=24000000] [depth=60] > gpurun_out/bam_e2e.json"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from finaletoolkit_amd import source, synth  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 24_000_000
depth = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
tmp = tempfile.mkdtemp(prefix="ftk_bam_")
path = os.path.join(tmp, "slice.bam")
t0 = time.time()
exp = synth.write_paired_bam(path, "mid", size, depth, 31)
res = {"slice_bp": size, "depth": depth, "pairs": exp["n"], "records": 2 * exp["n"], "file_MB": round(exp["file_bytes"] / 1e6, 1),
       "write_s": round(time.time() - t0, 1), "threads": source.usable_cores(), "reps": []}
ws, we = synth.tiling_windows(size, 100_000)
want_cov = None
for rep in range(4):
    source.close_all()
    eng = source.get_engine()
    t0 = time.perf_counter()
    for src, c in source.stream_source(path):
        t1 = time.perf_counter()
        key = src.key(c)
        r = eng.window_features(key, ws, we, 30, hist=(0, 1001), delfi=dict(quality_threshold=30))
        t2 = time.perf_counter()
        w = eng.wps(key, 0, size, size)
        t3 = time.perf_counter()
    cov = int(r["coverage"].sum())
    want_cov = cov if want_cov is None else want_cov
    assert cov == want_cov and len(w) == size
    st = src.decode_stage_ms
    total = t3 - t0
    res["reps"].append({"total_s": round(total, 4), "decode_until_resident_s": round(t1 - t0, 4), "features_s": round(t2 - t1, 4),
                        "wps_and_copy_back_s": round(t3 - t2, 4), "decoder_producer_stage_ms": st,
                        "inflate_share_of_total": round(st["inflate"] / 1e3 / total, 3) if st else None,
                        "fragments_per_s_M": round(exp["n"] / total / 1e6, 1), "windows_per_s": round(len(ws) / total, 1)})
    del w, r
source.close_all()
print(json.dumps(res))
