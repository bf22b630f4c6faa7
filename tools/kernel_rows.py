"""Rooflines of the kernels BEHIND the headline launch, measured live with HIP events on the stream they are launched on:

  next_rows    the §8(f) kernels on a chr2-sized 30x contig - ``cleavage_kernel`` (whole contig, float64 per base), the
               end-motif mode of the window-feature kernels (k = 4, 2bit reference, 1 Mb windows), ``adjust_median_kernel``
               (10 000 x 5 kb score runs, W = 1000), ``gc_count_kernel`` (100 kb bins of a 2bit image);
  bam_kernels  the window-feature blocks, the WPS tiles and the merged launch on a chr1-sized 60x contig WITH read1 columns
               (the BAM fetch rule, io/alignment.py:242-268): the kernels of BASELINE config 5.

``bench.py`` calls ``measure()`` for its ``next_rows`` / ``roofline.bam_kernels`` entries; run as a script (under
``rocprofv3 --kernel-trace --stats`` or ``--pmc`` for the summaries in profiles/) it prints the same dict as JSON.
usage: python3 tools/kernel_rows.py [next|bam|all] [reps=5]

Algorithmic bytes (SURVEY section 8-d's convention: what the result needs, once): fragment columns 10 B per fragment
(+8 B with read1 columns), 8 B per per-base score, outputs once; stated per row in ``algorithmic``.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0


def _row(kernel, ms, nbytes, algorithmic, launches=1, note=None):
    ms = np.asarray(ms, dtype=np.float64)
    med = float(np.median(ms))
    out = dict(kernel=kernel, achieved=round(nbytes / (med * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS, unit="GB/s",
               frac=round(nbytes / (med * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), algorithmic_bytes=int(nbytes),
               algorithmic=algorithmic, avg_launch_ms=round(med / launches, 4), best_launch_ms=round(float(ms.min()) / launches, 4),
               launches_timed=int(len(ms) * launches))
    if note:
        out["note"] = note
    return out


def _time(eng, fn, reps, flush=None):
    fn()
    eng.sync()
    ts = []
    for _ in range(reps):
        if flush is not None:
            flush()
        eng.event_record(10)
        fn()
        eng.event_record(11)
        ts.append(eng.event_elapsed_ms(10, 11))
    return ts


def measure(torch, eng, which="all", reps=5, seed=1):
    """``eng``: an Engine whose stream is torch's current stream.  Returns ``{"next_rows": {...}, "bam_kernels": {...}}``."""
    import ctypes as C
    from finaletoolkit_amd import _lib as L, synth
    dev = torch.device("cuda", torch.cuda.current_device())
    out = {}
    flush_buf = torch.empty(160_000_000, dtype=torch.int32, device=dev)  # 640 MB > the 256 MB Infinity Cache

    def flush():
        flush_buf.sum()  # evicts with clean lines: the kernel reads its columns from HBM

    if which in ("all", "next"):
        size = synth.B37_SIZES["2"]
        n = synth.n_fragments(size, 30.0)
        s, e, q, st = synth.gen_contig_device(torch, dev, size, n, seed)
        torch.cuda.synchronize()
        eng.load_contig_device("kr_next", s, e, q, st, n)
        rows = {}
        # cleavage profile of the whole contig into device memory
        cl = torch.empty(size, dtype=torch.float64, device=dev)
        f = lambda: eng.cleavage("kr_next", 0, size, None, None, 20, out=cl)
        rows["cleavage_kernel"] = _row("cleavage_kernel", _time(eng, f, reps, flush), 10 * n + 8 * size,
                                       "10 B x fragments + 8 B x bases (float64 per base)")
        del cl
        # end motifs, k = 4, 2bit reference, 1 Mb windows (end_motifs' tiling)
        rng = np.random.default_rng(5)
        packed = rng.integers(0, 256, (size + 3) // 4, dtype=np.uint8)
        rid = eng.ref_upload(("kr", "2bit"), packed, 1)
        eng.ref_set_layout(rid, size, 0, 0, [10_000], [20_000])
        mws, mwe = synth.tiling_windows(size, 1_000_000)
        k = 4
        d_counts = torch.zeros((len(mws), 4 ** k), dtype=torch.int32, device=dev)  # (the counts stay on the device: no host round trip inside the events)
        d_nfrag = torch.zeros(len(mws), dtype=torch.int64, device=dev)
        d_err = torch.zeros(len(mws), dtype=torch.int64, device=dev)
        mot = L.Motif(k, 0, -k, 1, 0, 0, 0)
        # windows resident on the device like the fragments and the image (the headline step's are too); the same call with
        # HOST window arrays - one small upload in front of the planner - is timed beside it
        d_mws, d_mwe = torch.from_numpy(mws).to(dev), torch.from_numpy(mwe).to(dev)
        f_host = lambda: eng._check(eng.lib.ftk_motif_counts(eng.ctx, eng.contig_id("kr_next"), rid, L.ptr(mws), L.ptr(mwe), len(mws),
                                                             C.byref(mot), 30, L.FETCH_TABIX, L.ptr(d_counts), L.ptr(d_nfrag), L.ptr(d_err)))
        f = lambda: eng._check(eng.lib.ftk_motif_counts(eng.ctx, eng.contig_id("kr_next"), rid, L.ptr(d_mws), L.ptr(d_mwe), len(mws),
                                                        C.byref(mot), 30, L.FETCH_TABIX, L.ptr(d_counts), L.ptr(d_nfrag), L.ptr(d_err)))
        ms_host = np.asarray(_time(eng, f_host, reps, flush))
        rows["motif_pass"] = _row("feat_*_kernel<CH=3> (end motifs k=4, 2bit)", _time(eng, f, reps, flush),
                                  10 * n + 2 * n + len(mws) * (4 ** k) * 4,
                                  "10 B x fragments + 2 x 1 B of packed reference per fragment + 4^k x 4 B per window",
                                  note="the planner, the window kernels and the count rows written to HBM; the window arrays are resident on the device, "
                                       "nothing crosses to or from the host inside the events")
        rows["motif_pass"]["with_host_window_arrays_ms"] = round(float(np.median(ms_host)), 4)
        got_host = eng.motif_counts("kr_next", rid, mws, mwe, k, 0, -k, True, False, 0, False, 30)[0]
        rows["motif_pass"]["device_counts_equal_host_call"] = bool(np.array_equal(d_counts.cpu().numpy().view(np.uint32), got_host))
        del d_counts, d_nfrag, d_err, d_mws, d_mwe
        # G + C of 100 kb bins from the 2bit image
        glo, ghi = synth.tiling_windows(size, 100_000)
        d_lo = torch.from_numpy(glo.astype(np.int64)).to(dev)
        d_hi = torch.from_numpy(ghi.astype(np.int64)).to(dev)
        d_gc = torch.zeros(len(glo), dtype=torch.int64, device=dev)
        f = lambda: eng._check(eng.lib.ftk_ref_gc_counts(eng.ctx, rid, L.ptr(d_lo), L.ptr(d_hi), len(glo), L.ptr(d_gc)))
        rows["gc_count_kernel"] = _row("gc_count_kernel", _time(eng, f, reps, flush), size // 4 + 8 * len(glo),
                                       "1/4 B per base (2bit image) + 8 B per bin",
                                       note="61 MB per launch: a plain sum-everything read of the same bytes takes 13.3-15.0 us by events on this device in "
                                            "every shape tried, an empty launch 6.2 us (tools/native/read_probe.hip, profiles/r6_read_probe.txt)")
        # raw WPS of the contig's first 50 Mb (the scores adjust_wps is made for): kept for the adjust row below
        n_iv, ilen, W = 10_000, 5_000, 1000
        wps_i64 = torch.empty(n_iv * ilen, dtype=torch.int64, device=dev)
        eng.wps("kr_next", 0, n_iv * ilen, size, out=wps_i64)
        x_wps = wps_i64.to(torch.float64)
        del wps_i64
        eng.release("kr_next")
        del s, e, q, st, d_lo, d_hi, d_gc
        # adjust_wps: running median W = 1000 over 10 000 x 5 kb score runs (device resident)
        n_iv, ilen, W = 10_000, 5_000, 1000
        g = torch.Generator(device=dev)
        g.manual_seed(1)
        x = torch.randint(-60, 60, (n_iv * ilen,), device=dev, generator=g).to(torch.float64)
        offs = np.arange(n_iv + 1, dtype=np.int64) * ilen
        y = torch.empty(n_iv * (ilen - W), dtype=torch.float64, device=dev)
        f = lambda: eng.wps_adjust(x.data_ptr(), offs, W, out=y.data_ptr(), savgol=False)
        rows["adjust_median_kernel"] = _row("adjust_median_kernel", _time(eng, f, reps), 8 * n_iv * ilen + 8 * n_iv * (ilen - W),
                                            "8 B per input score + 8 B per output",
                                            note="integer scores (raw WPS): adjust_median_hist_kernel, one sliding histogram per lane; the sort kernel "
                                                 "behind it finds every interval answered and exits")
        f = lambda: eng.wps_adjust(x_wps.data_ptr(), offs, W, out=y.data_ptr(), savgol=False)
        rows["adjust_median_kernel_on_wps"] = _row("adjust_median_kernel (the contig's own WPS)", _time(eng, f, reps), 8 * n_iv * ilen + 8 * n_iv * (ilen - W),
                                                   "8 B per input score + 8 B per output",
                                                   note="the scores the filter is made for: WPS (window 120, fragments 120-180) of the first 50 Mb of the 30x contig, "
                                                        "as 10 000 runs of 5 kb; value range %d..%d.  The row above is the same shape on uniformly random integers in "
                                                        "[-60, 60): every step replaces a value and moves the middle" % (int(x_wps.min().item()), int(x_wps.max().item())))
        del x_wps
        xf = x + 0.25  # the same runs as non-integers: every interval goes through the sort kernel (the path of round 5)
        f = lambda: eng.wps_adjust(xf.data_ptr(), offs, W, out=y.data_ptr(), savgol=False)
        rows["adjust_median_kernel_sort_path"] = _row("adjust_median_kernel (non-integer scores)", _time(eng, f, reps), 8 * n_iv * ilen + 8 * n_iv * (ilen - W),
                                                      "8 B per input score + 8 B per output",
                                                      note="an exact sliding median of arbitrary doubles: bound by its LDS sort and slot walk, not by HBM")
        del xf
        del x, y
        out["next_rows"] = rows
    if which in ("all", "bam"):
        size = synth.B37_SIZES["1"]
        n = synth.n_fragments(size, 60.0)
        s, e, q, st = synth.gen_contig_device(torch, dev, size, n, seed)
        e = torch.maximum(e, s + 50)
        r1s = torch.where(st == 1, s, e - 50).to(torch.int32).contiguous()
        r1e = (r1s + 50).contiguous()
        torch.cuda.synchronize()
        eng.load_contig_device("kr_bam", s, e, q, st, n)
        eng.set_read1("kr_bam", r1s, r1e, n)
        ws, we = synth.tiling_windows(size, 100_000)
        nw = len(ws)
        cov = torch.zeros(nw, dtype=torch.int64, device=dev)
        hist = torch.zeros((nw, 1001), dtype=torch.int32, device=dev)
        over = torch.zeros(nw, dtype=torch.int64, device=dev)
        sh = torch.zeros(nw, dtype=torch.int64, device=dev)
        lg = torch.zeros(nw, dtype=torch.int64, device=dev)
        w = torch.empty(size, dtype=torch.int64, device=dev)
        bl_s, bl_e = synth.synth_blacklist(size, 5, 160)
        gaps = synth.synth_gaps(size)
        flt = L.make_filter(30, None, None, "midpoint", L.FETCH_BAM_READ1)
        gp = L.make_gaps(gaps)
        feat_out = nw * (1001 * 4 + 32)
        f_feat = lambda: eng._check(eng.lib.ftk_window_features(
            eng.ctx, eng.contig_id("kr_bam"), L.ptr(ws), L.ptr(we), nw, C.byref(flt), L.ptr(cov), 0, 1001, L.ptr(hist),
            L.ptr(over), 30, L.ptr(bl_s), L.ptr(bl_e), len(bl_s), C.byref(gp), L.ptr(sh), L.ptr(lg)))
        f_wps = lambda: eng.wps("kr_bam", 0, size, size, 120, 120, 180, 30, out=w)
        f_one = lambda: eng.window_features_wps("kr_bam", ws, we, w, 0, size, size, coverage=cov, hist=hist, hist_bins=(0, 1001),
                                                overflow=over, delfi_q=30, bl_start=bl_s, bl_end=bl_e, gaps=gaps, short=sh, long=lg)
        # The read1 columns (8 B per fragment) are read only by the groups of four fragments that hold a fragment crossing
        # a window bound (ContigView::r1_inside; WPS: a fetch bound of the one interval) - the launch is priced at the bytes
        # it READS: 10 B per fragment + 8 B for the fragments of those groups.  The share is counted here, on the columns:
        # a fragment crosses a bound when its first and last base lie in different 100 kb windows; its whole group loads.
        crossing = int(((s // 100_000) != ((e - 1) // 100_000)).sum().item())
        r1_share = min(1.0, 4.0 * crossing / n)
        per_frag = 10.0 + 8.0 * r1_share
        note = (f"priced at the bytes the launch reads: 10 B per fragment + the 8 B read1 columns for the groups of four that hold "
                f"a fragment crossing a window bound ({crossing} of {n} fragments cross one: {per_frag:.3f} B per fragment; the WPS "
                "tiles of one whole-contig interval read none, priced the same).  frac_at_survey_bytes is the same launch at SURVEY "
                "8-d's 18 B per fragment - the columns a BAM contig HOLDS, most of which are never read, so it is not a roofline "
                "fraction and may exceed 1")
        rows = {}
        fb = int(per_frag * n)
        for key, kern, fn, b18, bread in (
                ("window_features", "feat_fast_kernel<512,1,1,1,BAM>", f_feat, 18 * n + 8 * nw + feat_out, fb + 8 * nw + feat_out),
                ("wps", "wps_stream_kernel", f_wps, 18 * n + 8 * size, fb + 8 * size),
                ("features_then_wps_one_launch", "feat_then_wps_kernel<1,1,1,BAM,NT>", f_one,
                 2 * 18 * n + 8 * size + 8 * nw + feat_out, 2 * fb + 8 * size + 8 * nw + feat_out)):
            ts = _time(eng, fn, reps, flush)
            r = _row(kern, ts, bread, f"{per_frag:.3f} B x fragments (x2 in the merged launch: feature blocks, then WPS tiles) + 8 B x bases + outputs",
                     note=note)
            r["achieved_at_survey_bytes"] = round(b18 / (float(np.median(ts)) * 1e-3) / 1e9, 1)
            r["frac_at_survey_bytes"] = round(r["achieved_at_survey_bytes"] / HBM_PEAK_GBS, 4)
            rows[key] = r
        rows["workload"] = f"chr1-sized contig, 60x, {n} fragments with read1 columns, {nw} x 100 kb windows, {size} bases"
        eng.release("kr_bam")
        out["bam_kernels"] = rows
    del flush_buf
    torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    import torch
    from finaletoolkit_amd.engine import Engine
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    eng = Engine(0)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    eng.set_stream(stream.cuda_stream)
    print(json.dumps(measure(torch, eng, which, reps)))
