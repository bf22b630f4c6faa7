#!/usr/bin/env python3
"""Device inflate alone: a contig's fragment text as BGZF (library writer, levels 1 and 6; bgzip-like zlib level 6
for a part) through ftk_bgzf_inflate_device.  Wall time here includes pageable H2D / D2H copies of the whole image;
run under `rocprofv3 --kernel-trace --stats` for the kernels' own durations.
usage: tools/inflate_bench.py [contig=21]"""
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from finaletoolkit_amd import _lib as L, bgzf, synth, writers  # noqa: E402
from finaletoolkit_amd.engine import Engine  # noqa: E402

contig = sys.argv[1] if len(sys.argv) > 1 else "21"
NOCHECK = os.environ.get("INFLATE_BENCH_NOCHECK") == "1"  # (timing builds that skip work: FTK_LANES_SKIP)
size = synth.B37_SIZES[contig]
s, e, q, st = synth.synth_contig(size, 30.0, 5)
eng = Engine(0)
tmp = tempfile.mkdtemp(prefix="ftk_inf_")
with writers.frag_rows(contig, s, e, q, st) as rows:
    text = rows.tobytes()
for level in (1, 6):
    p = os.path.join(tmp, f"l{level}.gz")
    writers.bgzf_write(p, text, level)
    image = open(p, "rb").read()
    out = np.zeros(len(text), np.uint8)
    n = C.c_int64()
    for rep in range(3):
        t0 = time.perf_counter()
        rc = eng.lib.ftk_bgzf_inflate_device(eng.ctx, image, len(image), L.ptr(out), len(out), C.byref(n))
        dt = time.perf_counter() - t0
        assert rc == 0 or NOCHECK, eng.lib.ftk_last_error(eng.ctx)
        print(f"libdeflate level {level}: {len(image) / 1e6:.1f} MB -> {n.value / 1e6:.1f} MB, {len(image) // 1 and -(-len(text) // 0xFF00)} blocks, "
              f"call {dt * 1e3:.1f} ms ({n.value / dt / 1e9:.1f} GB/s of text incl. copies)", flush=True)
    assert NOCHECK or out.tobytes() == text
# bgzip-like: zlib level 6 blocks (Python zlib, first 40 MB)
part = text[:40_000_000]
p = os.path.join(tmp, "z6.gz")
bgzf.write_bgzf(p, part, level=6)
image = open(p, "rb").read()
out = np.zeros(len(part), np.uint8)
n = C.c_int64()
for rep in range(3):
    t0 = time.perf_counter()
    rc = eng.lib.ftk_bgzf_inflate_device(eng.ctx, image, len(image), L.ptr(out), len(out), C.byref(n))
    dt = time.perf_counter() - t0
    assert rc == 0 or NOCHECK, eng.lib.ftk_last_error(eng.ctx)
    print(f"zlib level 6: {len(image) / 1e6:.1f} MB -> {n.value / 1e6:.1f} MB, call {dt * 1e3:.1f} ms", flush=True)
assert NOCHECK or out.tobytes() == part
