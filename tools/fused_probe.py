import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
import torch, bench
from finaletoolkit_amd import synth
from finaletoolkit_amd.engine import Engine
size = synth.B37_SIZES["2"]; dev = torch.device("cuda", 0); eng = Engine(0)
st_ = torch.cuda.Stream(); torch.cuda.set_stream(st_); eng.set_stream(st_.cuda_stream)
n = synth.n_fragments(size, 30.0)
s, e, q, st = bench.gen_contig_device(torch, dev, size, n, 1); torch.cuda.synchronize()
eng.load_contig_device("c", s, e, q, st, n)
nw = -(-size // 100_000)
out = torch.empty(size, dtype=torch.int64, device=dev)
cov = torch.zeros(nw, dtype=torch.int64, device=dev); over = torch.zeros(nw, dtype=torch.int64, device=dev)
hist = torch.zeros((nw, 1001), dtype=torch.int32, device=dev)
sh = torch.zeros(nw, dtype=torch.int64, device=dev); lg = torch.zeros(nw, dtype=torch.int64, device=dev)
def t(fn, name):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(10):
        eng.event_record(0); fn(); eng.event_record(1); ts.append(eng.event_elapsed_ms(0, 1))
    print(f"{name:40s} {np.median(ts)*1e3:8.1f} us")
t(lambda: eng.wps("c", 0, size, size, out=out), "wps alone")
t(lambda: eng.wps_window_features("c", size, 0, 100_000, nw, wps_out=out, coverage=cov), "fused: coverage")
t(lambda: eng.wps_window_features("c", size, 0, 100_000, nw, wps_out=out, short=sh, long=lg), "fused: delfi")
t(lambda: eng.wps_window_features("c", size, 0, 100_000, nw, wps_out=out, coverage=cov, short=sh, long=lg), "fused: coverage+delfi")
t(lambda: eng.wps_window_features("c", size, 0, 100_000, nw, wps_out=out, coverage=cov, hist=hist, hist_bins=(0, 1001), overflow=over), "fused: coverage+hist")
t(lambda: eng.wps_window_features("c", size, 0, 100_000, nw, wps_out=out, coverage=cov, hist=hist, hist_bins=(0, 64), overflow=over), "fused: coverage+hist64")
