#!/usr/bin/env python3
"""End-to-end stage timings on one contig: synthetic frag.gz on disk -> BGZF inflate + parse
(host threads) -> page-locked SoA -> H2D -> fused window features + WPS -> results on the host.
usage: tools/e2e_bench.py [contig] [threads]"""
import ctypes as C
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


from finaletoolkit_amd import _lib as L, bgzf, synth  # noqa: E402
from finaletoolkit_amd.engine import Engine  # noqa: E402

contig = sys.argv[1] if len(sys.argv) > 1 else "22"
threads = int(sys.argv[2]) if len(sys.argv) > 2 else source_threads()
size = synth.B37_SIZES[contig]
s, e, q, st = synth.synth_contig(size, 30.0, synth.SEED_BASE + 21)
n = len(s)
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "c.frag.gz")
t0 = time.time()
import pandas as pd  # noqa: E402
import io  # noqa: E402
buf = io.StringIO()
pd.DataFrame({"c": contig, "s": s, "e": e, "q": q, "t": np.where(st == 1, "+", "-")}).to_csv(
    buf, sep="\t", header=False, index=False)
text = buf.getvalue().encode()
bgzf.write_bgzf(path, text, level=6)
t_write = time.time() - t0
lib = L.load()
eng = Engine(0)
res = {"contig": contig, "fragments": n, "text_MB": round(len(text) / 1e6, 1),
       "file_MB": round(os.path.getsize(path) / 1e6, 1), "threads": threads, "write_s": round(t_write, 2)}
w = None
for rep in range(4):  # later repetitions = page cache warm, allocators warm, result block recycled
    t0 = time.perf_counter()
    table = C.c_void_p()
    assert lib.ftk_fragfile_decode(path.encode(), None, threads, C.byref(table)) == 0
    t1 = time.perf_counter()
    pinned = lib.ftk_fragtable_is_pinned(table, 0)
    eng.load_contig_from_table("c", table, 0, False)
    eng.sync()
    t2 = time.perf_counter()
    ws, we = synth.tiling_windows(size, 100_000)
    r = eng.window_features("c", ws, we, 30, hist=(0, 1001), delfi=dict(quality_threshold=30))
    t3 = time.perf_counter()
    w = None  # the previous result goes back to the library's page-locked cache
    w = eng.wps("c", 0, size, size)
    t4 = time.perf_counter()
    lib.ftk_fragtable_free(table)
    res[f"rep{rep}"] = {"decode_s": round(t1 - t0, 4), "decode_MBps_text": round(len(text) / 1e6 / (t1 - t0), 1),
                        "decode_Mfrag_s": round(n / 1e6 / (t1 - t0), 2), "pinned": int(pinned),
                        "upload_s": round(t2 - t1, 4), "upload_GBps": round(10 * n / 1e9 / (t2 - t1), 2),
                        "features_to_host_s": round(t3 - t2, 4), "wps_to_host_s": round(t4 - t3, 4),
                        "wps_D2H_GBps": round(8 * size / 1e9 / (t4 - t3), 2),
                        "end_to_end_s": round(t4 - t0, 4), "windows_per_s_end_to_end": round(len(ws) / (t4 - t0), 1),
                        "windows_per_s_without_wps_D2H": round(len(ws) / (t3 - t0), 1)}
assert int(r["coverage"].sum()) == int((q >= 30).sum())
print(json.dumps(res))
