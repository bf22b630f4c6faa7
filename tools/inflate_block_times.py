#!/usr/bin/env python3
"""Per-block decode times of the device inflate.  Needs the library built with -DFTK_INFLATE_TIMING:
    cd finaletoolkit_amd/csrc && hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -shared \
        -DFTK_INFLATE_TIMING -o ../libftk_timing.so *.hip *.cpp -lz -lpthread -ldl
    FTK_LIB=finaletoolkit_amd/libftk_timing.so python tools/inflate_block_times.py [slice_bp=2400000]
Prints, for a 60x BAM slice and for the same fragments as text, the distribution of the blocks' wavefront
lifetimes and when they started, relative to the first."""
import ctypes as C
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from finaletoolkit_amd import synth, writers, _lib as L  # noqa: E402
from finaletoolkit_amd.engine import Engine  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 2_400_000
d = tempfile.mkdtemp()
synth.write_paired_bam(d + "/x.bam", "mid", size, 60.0, 31)
eng = Engine(0)
s, e, q, st = synth.synth_contig(20_000_000, 30.0, 5)
with writers.frag_rows("20", s, e, q, st) as rows:
    text = rows.tobytes()
writers.bgzf_write(d + "/t.gz", text, 6)


def run(path, label):
    image = open(path, "rb").read()
    n = C.c_int64()
    one = np.zeros(1, np.uint8)
    eng.lib.ftk_bgzf_inflate_device(eng.ctx, image, len(image), L.ptr(one), 0, C.byref(n))
    out = np.zeros(n.value, np.uint8)
    for _ in range(2):
        rc = eng.lib.ftk_bgzf_inflate_device(eng.ctx, image, len(image), L.ptr(out), len(out), C.byref(n))
        assert rc == 0
    # count blocks: gzip members (BSIZE at offset 16 of each header)
    nb, off = 0, 0
    sizes = []
    while off < len(image):
        bs = int.from_bytes(image[off + 16:off + 18], "little") + 1
        sizes.append(bs)
        off += bs
        nb += 1
    nb = min(nb, 65536)
    ticks = np.zeros(2 * nb, np.uint64)
    eng.lib.ftk_debug_inflate_ticks.argtypes = [C.c_void_p, C.c_int]
    assert eng.lib.ftk_debug_inflate_ticks(ticks.ctypes.data, nb) == 0
    t = ticks.reshape(-1, 2).astype(np.int64)
    t0 = t[:, 0].min()
    start = (t[:, 0] - t0) / 100.0  # us
    dur = (t[:, 1] - t[:, 0]) / 100.0
    end = (t[:, 1] - t0) / 100.0
    pc = np.percentile(dur, [0, 10, 50, 90, 99, 100])
    print(f"{label}: {len(image) / 1e6:.0f} MB -> {n.value / 1e6:.0f} MB, {nb} blocks; kernel span {end.max():.0f} us")
    print("  block lifetime us  min/p10/p50/p90/p99/max:", " ".join(f"{x:.0f}" for x in pc))
    print("  start offset us    p50/p90/max:", " ".join(f"{x:.0f}" for x in np.percentile(start, [50, 90, 100])))
    worst = np.argsort(dur)[-5:][::-1]
    print("  slowest blocks:", [(int(k), int(dur[k]), sizes[k]) for k in worst], "(index, us, compressed bytes)")
    sz = np.array(sizes[:nb])
    print(f"  lifetime vs compressed size: corr {np.corrcoef(sz, dur)[0, 1]:.2f}; ns per compressed byte p50 {np.median(dur * 1e3 / sz):.0f}")


run(d + "/x.bam", "BAM")
run(d + "/t.gz", "text")
