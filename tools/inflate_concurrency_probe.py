#!/usr/bin/env python3
"""Do inflate launches on different HIP streams overlap on the device?  A 2.4 Mb slice of a 60x BAM (~55 MB, ~850
blocks: a quarter of the wavefronts the chip holds) through ftk_bgzf_inflate_device on three contexts (one stream
each), one call after the other and then from three threads at once.  Run under
`rocprofv3 --kernel-trace --output-format csv` to see the kernels' intervals."""
import ctypes as C
import os
import sys
import tempfile
import threading
import time

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from finaletoolkit_amd import synth, _lib as L  # noqa: E402
from finaletoolkit_amd.engine import Engine  # noqa: E402

d = tempfile.mkdtemp()
synth.write_paired_bam(d + "/x.bam", "mid", 2_400_000, 60.0, 31)
image = open(d + "/x.bam", "rb").read()
engs = [Engine(0) for _ in range(3)]
n = C.c_int64()
one = np.zeros(1, np.uint8)
engs[0].lib.ftk_bgzf_inflate_device(engs[0].ctx, image, len(image), L.ptr(one), 0, C.byref(n))
outs = [np.zeros(n.value, np.uint8) for _ in engs]


def call(k):
    m = C.c_int64()
    rc = engs[k].lib.ftk_bgzf_inflate_device(engs[k].ctx, image, len(image), L.ptr(outs[k]), len(outs[k]), C.byref(m))
    assert rc == 0


for k in range(3):
    call(k)  # warm
t = time.perf_counter()
for k in range(3):
    call(k)
seq = time.perf_counter() - t
th = [threading.Thread(target=call, args=(k,)) for k in range(3)]
t = time.perf_counter()
for x in th:
    x.start()
for x in th:
    x.join()
par = time.perf_counter() - t
print(f"image {len(image) / 1e6:.0f} MB -> {n.value / 1e6:.0f} MB; three calls in a row {seq * 1e3:.1f} ms, from three threads {par * 1e3:.1f} ms")
