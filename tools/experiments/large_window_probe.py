"""How the window kernels fare on FEW, LARGE windows (1 Mb tiles of a chr2-sized 30x contig): coverage, coverage +
length histogram, and the motif pass, event-timed with the columns evicted first.  usage: python3 tools/experiments/large_window_probe.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from finaletoolkit_amd import _lib as L, synth  # noqa: E402
from finaletoolkit_amd.engine import Engine  # noqa: E402
from kernel_rows import _time  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
eng = Engine(0)
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
eng.set_stream(stream.cuda_stream)
size = synth.B37_SIZES["2"]
n = synth.n_fragments(size, 30.0)
s, e, q, st = synth.gen_contig_device(torch, dev, size, n, 1)
torch.cuda.synchronize()
eng.load_contig_device("c", s, e, q, st, n)
flush_buf = torch.empty(160_000_000, dtype=torch.int32, device=dev)
flush = lambda: flush_buf.sum()
for wsize in (1_000_000, 100_000):
    ws, we = synth.tiling_windows(size, wsize)
    d_out = torch.zeros(len(ws), dtype=torch.int64, device=dev)
    for name, f in (("coverage, midpoint (fast form)", lambda: eng.window_counts("c", ws, we, 30, out=d_out)),
                    ("coverage, any (general form)", lambda: eng.window_counts("c", ws, we, 30, intersect_policy="any", out=d_out))):
        try:
            ms = _time(eng, f, 5, flush)
            print(wsize, name, "median %.1f us  best %.1f us  (%.2f of 8 TB/s at 10 B/fragment)" % (np.median(ms) * 1e3, min(ms) * 1e3, 10 * n / (np.median(ms) * 1e-3) / 8e12))
        except Exception as ex:  # noqa: BLE001
            print(wsize, name, "failed:", ex)
