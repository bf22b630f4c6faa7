#!/usr/bin/env python3
"""End to end on a multi-contig frag.gz: whole-file decode then compute, against the streaming decoder
(contig k+1 is inflated / parsed on the host threads while contig k is uploaded, run through the fused
window features + WPS and copied back).  usage: tools/e2e_stream_bench.py [contigs=19,20,21,22] [threads]"""
import ctypes as C
import io
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def source_threads():
    from finaletoolkit_amd.source import usable_cores
    return usable_cores()


from finaletoolkit_amd import _lib as L, bgzf, source, synth  # noqa: E402

names = (sys.argv[1] if len(sys.argv) > 1 else "19,20,21,22").split(",")
threads = int(sys.argv[2]) if len(sys.argv) > 2 else source_threads()
import pandas as pd  # noqa: E402

tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "multi.frag.gz")
t0 = time.time()
parts, n_total, truth = [], 0, {}
for c in names:
    size = synth.B37_SIZES[c]
    s, e, q, st = synth.synth_contig(size, 30.0, synth.SEED_BASE + list(synth.B37_SIZES).index(c))
    buf = io.StringIO()
    pd.DataFrame({"c": c, "s": s, "e": e, "q": q, "t": np.where(st == 1, "+", "-")}).to_csv(
        buf, sep="\t", header=False, index=False)
    parts.append(buf.getvalue().encode())
    n_total += len(s)
    truth[c] = int((q >= 30).sum())
text = b"".join(parts)
del parts
bgzf.write_bgzf(path, text, level=1)
open(path + ".tbi", "wb").close()
res = {"contigs": names, "fragments": n_total, "text_MB": round(len(text) / 1e6, 1),
       "file_MB": round(os.path.getsize(path) / 1e6, 1), "threads": threads, "write_s": round(time.time() - t0, 1)}
del text
eng = source.get_engine()
lib = L.load()


def compute(key, c):
    size = synth.B37_SIZES[c]
    ws, we = synth.tiling_windows(size, 100_000)
    r = eng.window_features(key, ws, we, 30, hist=(0, 1001), delfi=dict(quality_threshold=30))
    w = eng.wps(key, 0, size, size)
    assert int(r["coverage"].sum()) == truth[c] and len(w) == size
    return len(ws)


for rep in range(2):
    # (a) whole file first, then the contigs one after the other
    t0 = time.perf_counter()
    table = C.c_void_p()
    assert lib.ftk_fragfile_decode(path.encode(), None, threads, C.byref(table)) == 0
    t_dec = time.perf_counter()
    for i, c in enumerate(names):
        eng.load_contig_from_table("seq:" + c, table, i, False)
    lib.ftk_fragtable_free(table)
    t_up = time.perf_counter()
    n_win = sum(compute("seq:" + c, c) for c in names)
    t_seq = time.perf_counter()
    for c in names:
        eng.release("seq:" + c)
    # (b) streamed
    source.close_all()
    eng = source.get_engine()
    t1 = time.perf_counter()
    marks = []
    for src, c in source.stream_source(path, threads):
        ta = time.perf_counter()
        compute(src.key(c), c)
        marks.append((c, round(ta - t1, 4), round(time.perf_counter() - t1, 4)))
    t_str = time.perf_counter()
    res[f"rep{rep}"] = {
        "whole_file": {"decode_s": round(t_dec - t0, 4), "upload_s": round(t_up - t_dec, 4),
                       "compute_to_host_s": round(t_seq - t_up, 4), "end_to_end_s": round(t_seq - t0, 4),
                       "windows_per_s": round(n_win / (t_seq - t0), 1)},
        "streamed": {"end_to_end_s": round(t_str - t1, 4), "windows_per_s": round(n_win / (t_str - t1), 1),
                     "contig_resident_at_s / results_on_host_at_s": marks},
        "speedup": round((t_seq - t0) / (t_str - t1), 3)}
    source.close_all()
    eng = source.get_engine()
print(json.dumps(res))
