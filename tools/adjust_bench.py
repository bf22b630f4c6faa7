#!/usr/bin/env python3
"""Times ftk_wps_adjust (device-resident scores) against the numpy/scipy statement of the reference's
filter on the host.  usage: tools/adjust_bench.py [n_intervals] [interval_len] [W]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from finaletoolkit_amd.engine import Engine  # noqa: E402

n_iv = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
ilen = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
dev = torch.device("cuda", 0)
eng = Engine(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
eng.set_stream(stream.cuda_stream)
g = torch.Generator(device=dev)
g.manual_seed(1)
x = torch.randint(-60, 60, (n_iv * ilen,), device=dev, generator=g).to(torch.float64)
offs = np.arange(n_iv + 1, dtype=np.int64) * ilen
out = torch.empty(n_iv * (ilen - W), dtype=torch.float64, device=dev)
for name, kw in (("median+savgol", {}), ("median", dict(savgol=False)), ("mean", dict(mean=True, savgol=False))):
    eng.wps_adjust(x.data_ptr(), offs, W, out=out.data_ptr(), **kw)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        eng.event_record(0)
        eng.wps_adjust(x.data_ptr(), offs, W, out=out.data_ptr(), **kw)
        eng.event_record(1)
        ts.append(eng.event_elapsed_ms(0, 1))
    ms = float(np.median(ts))
    print(f"{name:14s} {n_iv} x {ilen} (W={W}): {ms:8.2f} ms  {n_iv * (ilen - W) / ms / 1e3:8.1f} M outputs/s  "
          f"{n_iv / ms * 1e3:9.0f} intervals/s", flush=True)
# host statement on a few intervals
from numpy.lib.stride_tricks import sliding_window_view  # noqa: E402
from scipy.signal import savgol_filter  # noqa: E402
h = x[:8 * ilen].cpu().numpy().reshape(8, ilen)
t0 = time.perf_counter()
for r in h:
    run = np.median(sliding_window_view(r, W)[:ilen - W], axis=1)
    savgol_filter(r[W // 2:-(W // 2)] - run, 21, 2)
dt = (time.perf_counter() - t0) / 8
print(f"numpy/scipy host: {dt * 1e3:.1f} ms per interval -> {1 / dt:.1f} intervals/s (1 core)")
got = out[:ilen - W].cpu().numpy()
