#!/usr/bin/env python3
"""Kernel micro-bench on one synthetic contig (GPU box): times ftk_wps / window features with
HIP events, interleaved repetitions.  usage: tools/kbench.py [contig_len] [reps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
from finaletoolkit_amd import synth  # noqa: E402
from finaletoolkit_amd.engine import Engine  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else synth.B37_SIZES["2"]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
eng = Engine(0)
_stream = torch.cuda.Stream()  # one explicit stream for torch ops and ftk launches (handle 0 = "own stream")
torch.cuda.set_stream(_stream)
eng.set_stream(_stream.cuda_stream)
DEPTH = float(os.environ.get("KBENCH_DEPTH", "30"))
BAM = os.environ.get("KBENCH_BAM", "0") != "0"   # read1 columns beside the fragments: the BAM fetch rule (io/alignment.py:245)
n = synth.n_fragments(size, DEPTH)
s, e, q, st = bench.gen_contig_device(torch, dev, size, n, 1)
torch.cuda.synchronize()
eng.load_contig_device("c", s, e, q, st, n)
FRAG_BYTES = 10
if BAM:
    e = torch.maximum(e, s + 50)
    eng.load_contig_device("c", s, e, q, st, n)
    r1s = torch.where(st == 1, s, e - 50).to(torch.int32).contiguous()
    r1e = (r1s + 50).contiguous()
    torch.cuda.synchronize()
    eng.set_read1("c", r1s, r1e, n)
    FRAG_BYTES = 18  # +8 B per fragment for the read1 columns
    print(f"BAM contig: {n} fragments at {DEPTH}x, read1 columns resident; is_bam={eng.is_bam('c')}", flush=True)
ws, we = synth.tiling_windows(size, 100_000)
d_ws, d_we = torch.from_numpy(ws).to(dev), torch.from_numpy(we).to(dev)
out = torch.empty(size, dtype=torch.int64, device=dev)
cov = torch.zeros(len(ws), dtype=torch.int64, device=dev)
hist = torch.zeros((len(ws), 1001), dtype=torch.int32, device=dev)
over = torch.zeros(len(ws), dtype=torch.int64, device=dev)


flush_buf = torch.empty(160_000_000, dtype=torch.int32, device=dev)  # 640 MB > Infinity Cache


def timeit(fn, name, nbytes, cold=False):
    fn()
    torch.cuda.synchronize()
    ts = []
    if cold == "chain":  # 10 launches back to back between two events: no idle gap in which host preparation would count
        for _ in range(reps):
            flush_buf[:16_000_000].sum()  # keeps the GPU busy while the first launch is being prepared
            eng.event_record(0)
            for _ in range(10):
                fn()
            eng.event_record(1)
            ts.append(eng.event_elapsed_ms(0, 1) / 10)
        ts = np.array(ts)
        print(f"{name:28s} median {np.median(ts)*1e3:9.1f} us  min {ts.min()*1e3:9.1f} us   "
              f"{nbytes/np.median(ts)/1e6:8.1f} GB/s (median)  {nbytes/ts.min()/1e6:8.1f} GB/s (best)", flush=True)
        return
    for _ in range(reps):
        if cold == "read":
            flush_buf.sum()      # evict with clean lines
        elif cold:
            flush_buf.fill_(1)  # evict the contig from the 256 MiB Infinity Cache (dirty lines)
        eng.event_record(0)
        fn()
        eng.event_record(1)
        ts.append(eng.event_elapsed_ms(0, 1))
    ts = np.array(ts)
    print(f"{name:28s} median {np.median(ts)*1e3:9.1f} us  min {ts.min()*1e3:9.1f} us   "
          f"{nbytes/np.median(ts)/1e6:8.1f} GB/s (median)  {nbytes/ts.min()/1e6:8.1f} GB/s (best)", flush=True)


which = os.environ.get("KBENCH", "wps,cov,hist").split(",")
if "cal" in which:
    src = torch.ones(size, dtype=torch.int64, device=dev)
    timeit(lambda: out.fill_(7), "torch fill int64", 8 * size)
    timeit(lambda: out.copy_(src), "torch copy int64 (r+w)", 16 * size)
    del src
if "wps" in which:
    timeit(lambda: eng.wps("c", 0, size, size, 120, 120, 180, 30, out=out), "wps W=120 120-180", FRAG_BYTES * n + 8 * size)
if "merged" in which:
    # the step's launch: feature blocks (coverage + 1001-bin histogram + DELFI with blacklist and gaps) first, WPS tiles behind
    bl_s, bl_e = bench.synth_blacklist(size, 5, 160)
    sh = torch.zeros(len(ws), dtype=torch.int64, device=dev)
    lg = torch.zeros(len(ws), dtype=torch.int64, device=dev)
    gp = bench.synth_gaps(size)
    f = lambda: eng.window_features_wps("c", ws, we, out, 0, size, size, coverage=cov, hist=hist, hist_bins=(0, 1001),
                                        overflow=over, delfi_q=30, bl_start=bl_s, bl_end=bl_e, gaps=gp, short=sh, long=lg)
    nb = 2 * FRAG_BYTES * n + 8 * size + len(ws) * (1001 * 4 + 32)
    timeit(f, "features + WPS, one launch", nb)
    timeit(f, "features + WPS x10 chained", nb, cold="chain")
if "rd" in which:
    big = torch.ones(60_000_000, dtype=torch.int32, device=dev)  # 240 MB
    timeit(lambda: big.sum(), "torch sum 240MB warm", 240e6)
    timeit(lambda: big.sum(), "torch sum 240MB COLD", 240e6, cold=True)
    timeit(lambda: flush_buf.sum(), "torch sum 640MB (always cold)", 640e6)
    del big
if "cov" in which:
    timeit(lambda: eng.window_counts("c", d_ws, d_we, 30, out=cov), "window_counts 100kb", 10 * n)
    timeit(lambda: eng.window_counts("c", d_ws, d_we, 30, out=cov), "window_counts 100kb COLD", 10 * n, cold=True)
    timeit(lambda: eng.window_counts("c", d_ws, d_we, 30, out=cov), "window_counts COLD(read-flush)", 10 * n, cold="read")
if "hist" in which:
    import ctypes as C
    from finaletoolkit_amd import _lib as L
    flt = L.make_filter(30, None, None, "midpoint")
    timeit(lambda: eng._check(eng.lib.ftk_fraglen_hist(eng.ctx, eng.contig_id("c"), L.ptr(d_ws), L.ptr(d_we), len(ws),
                                                       C.byref(flt), 0, 1001, L.ptr(hist), L.ptr(over))),
           "fraglen_hist 100kb x1001", 10 * n)
    timeit(lambda: eng._check(eng.lib.ftk_fraglen_hist(eng.ctx, eng.contig_id("c"), L.ptr(d_ws), L.ptr(d_we), len(ws),
                                                       C.byref(flt), 0, 1001, L.ptr(hist), L.ptr(over))),
           "fraglen_hist COLD", 10 * n, cold=True)
if "feat" in which:
    import ctypes as C
    from finaletoolkit_amd import _lib as L
    flt = L.make_filter(30, None, None, "midpoint", L.FETCH_BAM_READ1 if BAM else L.FETCH_TABIX)
    sh = torch.zeros(len(ws), dtype=torch.int64, device=dev)
    lg = torch.zeros(len(ws), dtype=torch.int64, device=dev)
    bl_s, bl_e = bench.synth_blacklist(size, 5, 160)
    g = L.make_gaps(bench.synth_gaps(size))

    def fused(cov_on=True, hist_on=True, delfi_on=True):
        eng._check(eng.lib.ftk_window_features(
            eng.ctx, eng.contig_id("c"), L.ptr(ws), L.ptr(we), len(ws), C.byref(flt), L.ptr(cov) if cov_on else None,
            0, 1001, L.ptr(hist) if hist_on else None, L.ptr(over) if hist_on else None, 30, L.ptr(bl_s), L.ptr(bl_e),
            len(bl_s), C.byref(g), L.ptr(sh) if delfi_on else None, L.ptr(lg) if delfi_on else None))
    for name, kw in [("cov", dict(hist_on=False, delfi_on=False)), ("cov+hist", dict(delfi_on=False)),
                     ("delfi", dict(cov_on=False, hist_on=False)), ("cov+hist+delfi", {})]:
        timeit(lambda: fused(**kw), "fused " + name + " x10 chained", FRAG_BYTES * n, cold="chain")
        timeit(lambda: fused(**kw), "fused " + name + " COLD(read)", FRAG_BYTES * n, cold="read")
        timeit(lambda: fused(**kw), "fused " + name + " COLD(dirty)", FRAG_BYTES * n, cold=True)
if "cleave" in which:
    # whole-contig cleavage profile into a device buffer (float64 per base, like WPS's int64)
    import ctypes as C
    from finaletoolkit_amd import _lib as L
    cl = torch.empty(size, dtype=torch.float64, device=dev)
    s0 = np.array([0], np.int64); s1 = np.array([size], np.int64); so = np.array([0], np.int64)
    f = lambda: eng._check(eng.lib.ftk_cleavage_intervals(eng.ctx, eng.contig_id("c"), L.ptr(s0), L.ptr(s1), 1, L.ptr(so),
                                                          L.LEN_OPEN, L.LEN_OPEN, 20, L.ptr(cl)))
    timeit(f, "cleavage whole contig", 10 * n + 8 * size)
if "gc" in which:
    rng = np.random.default_rng(6)
    packed = rng.integers(0, 256, (size + 3) // 4, dtype=np.uint8)
    rid = eng.ref_upload(("kb", "gc2bit"), packed, 1)
    glo, ghi = synth.tiling_windows(size, 100_000)
    d_lo = torch.from_numpy(glo.astype(np.int64)).to(dev); d_hi = torch.from_numpy(ghi.astype(np.int64)).to(dev)
    d_gc = torch.zeros(len(glo), dtype=torch.int64, device=dev)
    import ctypes as C
    from finaletoolkit_amd import _lib as L
    f = lambda: eng._check(eng.lib.ftk_ref_gc_counts(eng.ctx, rid, L.ptr(d_lo), L.ptr(d_hi), len(glo), L.ptr(d_gc)))
    timeit(f, "GC count 100 kb bins (2bit)", size // 4)
if "small" in which:
    # window sizes from 500 bp to 100 kb over the same contig: which launch shape the host picks matters here
    for wlen in (500, 2_000, 10_000, 20_000, 100_000):
        sws, swe = synth.tiling_windows(size, wlen)
        timeit(lambda: eng.window_counts("c", sws, swe, 30), f"window_counts {wlen} bp x{len(sws)}", 10 * n)
if "motif" in which:
    # 1 Mb windows (end_motifs' tiling), random 2bit / FASTA-text images of the contig
    mws, mwe = synth.tiling_windows(size, 1_000_000)
    rng = np.random.default_rng(5)
    packed = rng.integers(0, 256, (size + 3) // 4, dtype=np.uint8)
    rid2 = eng.ref_upload(("kb", "2bit"), packed, 1)
    eng.ref_set_layout(rid2, size, 0, 0, [10_000], [20_000])
    text = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size + size // 60 + 1)].copy()
    text[60::61] = 10
    ridf = eng.ref_upload(("kb", "fa"), text, 0)
    eng.ref_set_layout(ridf, size, 60, 61)
    for rid, tag in ((rid2, "2bit"), (ridf, "fasta")):
        for k in (4, 6):
            f = lambda: eng.motif_counts("c", rid, mws, mwe, k, 0, -k, True, False, 0, False, 30)
            timeit(f, f"end motifs k={k} {tag}", 10 * n)
    c, nf, er = eng.motif_counts("c", rid2, mws, mwe, 4, 0, -4, True, False, 0, False, 30)
    print("motif total", int(c.sum()), "fragments", int(nf.sum()))
print("wps checksum", int(out[:5_000_000].sum().item()), "cov", int(cov.sum().item()))
