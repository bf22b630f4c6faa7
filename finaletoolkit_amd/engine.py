"""
``Engine``: one MI355X context holding fragments resident in HBM and running
the per-window kernels through the C ABI (``include/ftk.h``).

This is the host-side object the ``finaletoolkit_amd.frag`` functions drive.
It replaces the reference's "open the file per window, stream Python tuples"
feeder (``utils/_frag_generator.py:112-130``) by "decode once, keep the SoA in
HBM, answer every window in one launch".
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import numpy as np

from . import _lib as L


def _win(values, open_value):
    if isinstance(values, np.ndarray) and values.dtype == np.int32:
        return np.ascontiguousarray(values)
    if hasattr(values, "data_ptr"):  # int32 torch tensor (host or device), passed through by address
        return values
    return np.ascontiguousarray([open_value if v is None else int(v) for v in values], dtype=np.int32)


_SAVGOL = {}


def savgol_operators(window: int, deg: int):
    """Savitzky-Golay taps and the edge-fit matrices of ``scipy.signal.savgol_filter(mode="interp")``:
    ``coef`` [window] (interior correlation taps) and ``edge`` [2*half, window] whose first ``half`` rows
    map the first ``window`` samples to the first ``half`` outputs and whose last ``half`` rows map the last
    ``window`` samples to the last ``half`` outputs (polyfit + polyval is linear in the samples)."""
    key = (int(window), int(deg))
    if key not in _SAVGOL:
        from scipy.signal import savgol_coeffs
        if window % 2 == 0:
            raise ValueError("savgol_window_size must be odd")
        coef = np.ascontiguousarray(savgol_coeffs(window, deg), dtype=np.float64)
        half = window // 2
        x = np.arange(window)
        eye = np.eye(window)
        edge = np.zeros((2 * half, window))
        for j in range(window):
            p = np.polyfit(x, eye[j], deg)
            edge[:half, j] = np.polyval(p, np.arange(0, half))
            edge[half:, j] = np.polyval(p, np.arange(window - half, window))
        _SAVGOL[key] = (coef, np.ascontiguousarray(edge))
    return _SAVGOL[key]


class _HostBlock:
    """Owner of one host block of the library's result caches (``ftk_host_alloc``: page-locked;
    ``ftk_host_alloc_pageable``: ordinary memory); numpy arrays made from it keep it alive through
    ``__array_interface__`` and the block goes back to the library's cache with the last of them."""

    def __init__(self, lib, nbytes: int, pinned: bool = True):
        p = C.c_void_p()
        rc = (lib.ftk_host_alloc if pinned else lib.ftk_host_alloc_pageable)(int(nbytes), C.byref(p))
        if rc != 0:
            raise L.FtkError(rc, lib.ftk_fragtable_error().decode())
        self._lib, self._ptr = lib, p.value
        self.__array_interface__ = {"data": (p.value, False), "shape": (int(nbytes),), "typestr": "|u1", "version": 3}

    def __del__(self):
        if getattr(self, "_ptr", None):
            try:
                self._lib.ftk_host_free(self._ptr)
            except Exception:  # interpreter shutdown: the library object may already be gone
                pass
            self._ptr = None


# results of at least this many bytes are handed out in page-locked memory (their device -> host copy is the
# long leg of a per-base call); smaller ones are ordinary numpy arrays
PINNED_RESULT_MIN = 8 << 20
_PINNED_RESULTS = __import__("os").environ.get("FTK_PINNED_RESULTS", "1") != "0"
# ftk_wps sends a host output of this many positions or more across the link as int16 and lets the host threads widen
# it into the output (ftk_api.hip, kNarrowMin; FTK_WPS_NARROW_WIRE=0 keeps the plain copy): the device never writes
# such an output, so page-locking it buys nothing and costs 0.2 ms per MB in a process's first call
NARROW_WIRE_MIN = 1 << 22
_NARROW_WIRE = __import__("os").environ.get("FTK_WPS_NARROW_WIRE", "1") != "0"


class Engine:
    def __init__(self, device: int = 0):
        self.lib = L.load()
        ctx = C.c_void_p()
        rc = self.lib.ftk_ctx_create(int(device), C.byref(ctx))
        if rc != L.FTK_OK:
            raise L.FtkError(rc, self.lib.ftk_last_error(None).decode())
        self.ctx = ctx
        self.device = int(device)
        self._ids: dict[str, int] = {}
        self._next_id = 0
        self._bam: dict[str, bool] = {}
        # results whose device -> host copy may still be in flight (wps_async), by token: the engine keeps
        # the page-locked block alive until the copy is known to be done, whatever the caller drops
        self._pending: dict[int, np.ndarray] = {}

    # -- plumbing -------------------------------------------------------------
    def close(self):
        if getattr(self, "ctx", None):
            if self._pending:
                self.lib.ftk_ctx_sync(self.ctx)  # no DMA may outlive its target block
                self._pending.clear()
            self.lib.ftk_ctx_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def result_array(self, n: int, dtype, pinned: bool = True) -> np.ndarray:
        """Uninitialised result array of ``n`` elements: page-locked (``ftk_host_alloc``) when large, so that
        the copy back from the device is one DMA; the memory returns to the library's cache when the array
        (and every view of it) is gone.  ``pinned=False``: a recycled block of ordinary memory
        (``ftk_host_alloc_pageable``) for results the host threads fill."""
        dt = np.dtype(dtype)
        nbytes = int(n) * dt.itemsize
        if nbytes < PINNED_RESULT_MIN or not _PINNED_RESULTS:
            return np.empty(int(n), dt)
        try:
            return np.asarray(_HostBlock(self.lib, nbytes, pinned)).view(dt)
        except L.FtkError as e:
            if e.code != L.FTK_ERR_OOM:
                raise
            return np.empty(int(n), dt)  # the limit on page-locked results is reached: ordinary memory

    def _check(self, rc):
        if rc != L.FTK_OK:
            raise L.FtkError(rc, self.lib.ftk_last_error(self.ctx).decode())

    def set_stream(self, hip_stream: Optional[int]):
        self._check(self.lib.ftk_ctx_set_stream(self.ctx, C.c_void_p(hip_stream or 0)))

    def sync(self):
        self._check(self.lib.ftk_ctx_sync(self.ctx))
        self._pending.clear()

    def timer_start(self):
        self._check(self.lib.ftk_timer_start(self.ctx))

    def timer_stop(self) -> float:
        ms = C.c_float()
        self._check(self.lib.ftk_timer_stop(self.ctx, C.byref(ms)))
        return float(ms.value)

    def event_record(self, slot: int):
        self._check(self.lib.ftk_event_record(self.ctx, int(slot)))

    def event_elapsed_ms(self, slot_a: int, slot_b: int) -> float:
        ms = C.c_float()
        self._check(self.lib.ftk_event_elapsed_ms(self.ctx, int(slot_a), int(slot_b), C.byref(ms)))
        return float(ms.value)

    # -- fragments ------------------------------------------------------------
    def _new_id(self, name: str) -> int:
        """Contig id of ``name``; fresh ids are never reused (released ids must not alias live contigs)."""
        if name not in self._ids:
            self._ids[name] = self._next_id
            self._next_id += 1
        return self._ids[name]

    def contig_id(self, name: str) -> int:
        if name not in self._ids:
            raise KeyError(name)
        return self._ids[name]

    def has_contig(self, name: str) -> bool:
        return name in self._ids

    @property
    def contigs(self):
        return list(self._ids)

    def is_bam(self, name: str) -> bool:
        return self._bam.get(name, False)

    def load_contig(self, name: str, start, end, mapq, strand=None, r1_start=None, r1_end=None):
        """Upload one contig's start-sorted fragments (host numpy arrays)."""
        start = np.ascontiguousarray(start, dtype=np.int32)
        end = np.ascontiguousarray(end, dtype=np.int32)
        mapq = np.ascontiguousarray(mapq, dtype=np.uint8)
        strand = None if strand is None else np.ascontiguousarray(strand, dtype=np.uint8)
        n = len(start)
        if not (len(end) == n and len(mapq) == n and (strand is None or len(strand) == n)):
            raise ValueError("fragment columns differ in length")
        cid = self._new_id(name)
        self._check(self.lib.ftk_frags_from_host(self.ctx, cid, L.ptr(start), L.ptr(end), L.ptr(mapq),
                                                 L.ptr(strand), n))
        self._bam[name] = False
        if r1_start is not None:
            r1s = np.ascontiguousarray(r1_start, dtype=np.int32)
            r1e = np.ascontiguousarray(r1_end, dtype=np.int32)
            self._check(self.lib.ftk_frags_set_read1(self.ctx, cid, L.ptr(r1s), L.ptr(r1e), n))
            self._bam[name] = True
        return cid

    def load_contig_from_table(self, name: str, table, index: int, is_bam: bool):
        """Upload contig ``index`` of a decoded ``ftk_fragtable`` (page-locked columns -> HBM)."""
        cid = self._new_id(name)
        self._check(self.lib.ftk_frags_from_table(self.ctx, cid, table, int(index)))
        # an empty BAM contig carries no read1 columns: plain fetch mode is equivalent
        self._bam[name] = bool(is_bam) and self.lib.ftk_fragtable_contig_rows(table, int(index)) > 0
        return cid

    def load_contig_device(self, name: str, d_start, d_end, d_mapq, d_strand, n: int):
        """Adopt columns already in HBM (torch tensors or raw device addresses)."""
        cid = self._new_id(name)
        self._check(self.lib.ftk_frags_from_device(self.ctx, cid, L.ptr(d_start), L.ptr(d_end), L.ptr(d_mapq),
                                                   L.ptr(d_strand), int(n)))
        self._bam[name] = False
        return cid

    def set_read1(self, name: str, r1_start, r1_end, n: int):
        """Read1 spans of a resident contig's fragments (host arrays, torch tensors or raw device addresses): the contig
        then answers with the BAM fetch rule (io/alignment.py:245)."""
        self._check(self.lib.ftk_frags_set_read1(self.ctx, self.contig_id(name), L.ptr(r1_start), L.ptr(r1_end), int(n)))
        self._bam[name] = True

    def info(self, name: str):
        n, ml, me = C.c_int64(), C.c_int32(), C.c_int32()
        self._check(self.lib.ftk_frags_info(self.ctx, self.contig_id(name), C.byref(n), C.byref(ml), C.byref(me)))
        return int(n.value), int(ml.value), int(me.value)

    def release(self, name: str):
        self._check(self.lib.ftk_frags_release(self.ctx, self._ids.pop(name)))
        self._bam.pop(name, None)

    def _filter(self, name, quality_threshold, min_length, max_length, intersect_policy):
        return L.make_filter(quality_threshold, min_length, max_length, intersect_policy,
                             L.FETCH_BAM_READ1 if self.is_bam(name) else L.FETCH_TABIX)

    # -- features -------------------------------------------------------------
    def window_counts(self, name: str, starts: Sequence, stops: Sequence, quality_threshold=30, min_length=None,
                      max_length=None, intersect_policy="midpoint", out=None):
        """a5: fragments per window (frag/_coverage.py:117-130)."""
        ws, we = _win(starts, L.OPEN_LO), _win(stops, L.OPEN_HI)
        f = self._filter(name, quality_threshold, min_length, max_length, intersect_policy)
        res = np.zeros(len(ws), np.int64) if out is None else out
        self._check(self.lib.ftk_window_counts(self.ctx, self.contig_id(name), L.ptr(ws), L.ptr(we), len(ws),
                                               C.byref(f), L.ptr(res)))
        return res

    def delfi_counts(self, name: str, starts, stops, quality_threshold=30, bl_start=None, bl_end=None, gaps=None):
        """a10: DELFI short/long/num_frags per window (frag/_delfi.py:443-472)."""
        ws, we = _win(starts, L.OPEN_LO), _win(stops, L.OPEN_HI)
        n_bl = 0 if bl_start is None else len(bl_start)
        bs = None if n_bl == 0 else np.ascontiguousarray(bl_start, dtype=np.int32)
        be = None if n_bl == 0 else np.ascontiguousarray(bl_end, dtype=np.int32)
        g = L.make_gaps(gaps)
        sh = np.zeros(len(ws), np.int64)
        lg = np.zeros(len(ws), np.int64)
        nf = np.zeros(len(ws), np.int64)
        self._check(self.lib.ftk_delfi_counts(self.ctx, self.contig_id(name), L.ptr(ws), L.ptr(we), len(ws),
                                              int(quality_threshold), L.ptr(bs), L.ptr(be), n_bl, C.byref(g),
                                              L.ptr(sh), L.ptr(lg), L.ptr(nf)))
        return sh, lg, nf

    def fraglen_hist(self, name: str, starts, stops, len_lo: int, n_bins: int, quality_threshold=30,
                     min_length=None, max_length=None, intersect_policy="midpoint"):
        """a9: per-window length histogram (frag/_frag_length.py:147-153)."""
        ws, we = _win(starts, L.OPEN_LO), _win(stops, L.OPEN_HI)
        f = self._filter(name, quality_threshold, min_length, max_length, intersect_policy)
        hist = np.zeros((len(ws), int(n_bins)), np.uint32)
        over = np.zeros(len(ws), np.int64)
        self._check(self.lib.ftk_fraglen_hist(self.ctx, self.contig_id(name), L.ptr(ws), L.ptr(we), len(ws),
                                              C.byref(f), int(len_lo), int(n_bins), L.ptr(hist), L.ptr(over)))
        return hist, over

    def fraglen_stats(self, name: str, starts, stops, len_lo: int, n_bins: int, short_cut: int, quality_threshold=30,
                      min_length=None, max_length=None, intersect_policy="midpoint"):
        """a9: per-window length statistics computed on the device from the histogram rows (frag/_frag_length.py:156-172,
        202-224): float64 ``[n_win, 7]`` = mean median stdev min max count n_short; count 0 = no passing fragment."""
        ws, we = _win(starts, L.OPEN_LO), _win(stops, L.OPEN_HI)
        f = self._filter(name, quality_threshold, min_length, max_length, intersect_policy)
        out = np.zeros((len(ws), 7), np.float64)
        self._check(self.lib.ftk_fraglen_stats(self.ctx, self.contig_id(name), L.ptr(ws), L.ptr(we), len(ws), C.byref(f),
                                               int(len_lo), int(n_bins), int(short_cut), L.ptr(out)))
        return out

    def window_features(self, name: str, starts, stops, quality_threshold=30, min_length=None, max_length=None,
                        intersect_policy="midpoint", coverage=True, hist=None, delfi=None):
        """Coverage / length histogram / DELFI counts of the same windows in ONE pass
        (``ftk_window_features``).  ``hist=(len_lo, n_bins)``; ``delfi=dict(quality_threshold=,
        bl_start=, bl_end=, gaps=)``.  Returns a dict of the requested arrays."""
        ws, we = _win(starts, L.OPEN_LO), _win(stops, L.OPEN_HI)
        n = len(ws)
        f = self._filter(name, quality_threshold, min_length, max_length, intersect_policy)
        out = {}
        cov = np.zeros(n, np.int64) if coverage else None
        h = o = None
        len_lo = n_bins = 0
        if hist is not None:
            len_lo, n_bins = int(hist[0]), int(hist[1])
            h, o = np.zeros((n, n_bins), np.uint32), np.zeros(n, np.int64)
        sh = lg = bs = be = None
        n_bl, q, g = 0, 0, L.make_gaps(None)
        if delfi is not None:
            sh, lg = np.zeros(n, np.int64), np.zeros(n, np.int64)
            q = int(delfi.get("quality_threshold", 30))
            if delfi.get("bl_start") is not None and len(delfi["bl_start"]):
                bs = np.ascontiguousarray(delfi["bl_start"], dtype=np.int32)
                be = np.ascontiguousarray(delfi["bl_end"], dtype=np.int32)
                n_bl = len(bs)
            g = L.make_gaps(delfi.get("gaps"))
        self._check(self.lib.ftk_window_features(self.ctx, self.contig_id(name), L.ptr(ws), L.ptr(we), n, C.byref(f),
                                                 L.ptr(cov), len_lo, n_bins, L.ptr(h), L.ptr(o), q, L.ptr(bs),
                                                 L.ptr(be), n_bl, C.byref(g), L.ptr(sh), L.ptr(lg)))
        if coverage:
            out["coverage"] = cov
        if hist is not None:
            out["hist"], out["overflow"] = h, o
        if delfi is not None:
            out["short"], out["long"] = sh, lg
        return out

    def window_features_wps(self, name: str, starts, stops, wps_out, wps_start: int, wps_stop: int, chrom_size: int,
                            quality_threshold=30, min_length=None, max_length=None, intersect_policy="midpoint",
                            coverage=None, hist=None, hist_bins=None, overflow=None, delfi_q=30, bl_start=None,
                            bl_end=None, gaps=None, short=None, long=None, window_size=120, wps_min_length=120,
                            wps_max_length=180, wps_quality=30):
        """``window_features`` followed by ``wps`` of the same contig as ONE launch when the request allows it
        (``ftk_window_features_wps``: feature blocks first, WPS tiles behind them); every output is a device
        pointer / tensor or a host numpy array the caller allocated, ``wps_out`` must live on the device for the
        merged launch (a host array runs the two launches)."""
        ws, we = _win(starts, L.OPEN_LO), _win(stops, L.OPEN_HI)
        f = self._filter(name, quality_threshold, min_length, max_length, intersect_policy)
        len_lo, n_bins = hist_bins if hist_bins is not None else (0, 0)
        bs = be = None
        n_bl = 0
        if bl_start is not None and len(bl_start):
            bs = np.ascontiguousarray(bl_start, dtype=np.int32)
            be = np.ascontiguousarray(bl_end, dtype=np.int32)
            n_bl = len(bs)
        g = L.make_gaps(gaps)
        self._check(self.lib.ftk_window_features_wps(
            self.ctx, self.contig_id(name), L.ptr(ws), L.ptr(we), len(ws), C.byref(f), L.ptr(coverage), int(len_lo),
            int(n_bins), L.ptr(hist), L.ptr(overflow), int(delfi_q), L.ptr(bs), L.ptr(be), n_bl, C.byref(g), L.ptr(short),
            L.ptr(long), int(wps_start), int(wps_stop), int(chrom_size), int(window_size), int(wps_min_length),
            int(wps_max_length), int(wps_quality), L.ptr(wps_out)))

    def all_features_wps(self, name: str, starts, stops, chrom_size: int, quality_threshold=30, hist_bins=(0, 1001),
                         delfi_q=30, bl_start=None, bl_end=None, gaps=None, window_size=120, wps_min_length=120,
                         wps_max_length=180, wps_quality=30):
        """Every feature of the windows AND the WPS of every base of the contig, in host memory, from ONE launch
        (``window_features_wps``: coverage, length histogram + overflow, DELFI short / long; scores of
        ``[0, chrom_size)``) - BASELINE config 5's "all features fused single pass" as this engine serves it.
        Returns ``(features dict, scores int64)``."""
        ws, we = _win(starts, L.OPEN_LO), _win(stops, L.OPEN_HI)
        n = len(ws)
        f = dict(coverage=np.zeros(n, np.int64), hist=np.zeros((n, int(hist_bins[1])), np.uint32), overflow=np.zeros(n, np.int64),
                 short=np.zeros(n, np.int64), long=np.zeros(n, np.int64))
        size = int(chrom_size)
        w = self.result_array(size, np.int64, pinned=not (_NARROW_WIRE and size >= NARROW_WIRE_MIN))
        self.window_features_wps(name, ws, we, w, 0, size, size, quality_threshold, coverage=f["coverage"], hist=f["hist"],
                                 hist_bins=hist_bins, overflow=f["overflow"], delfi_q=delfi_q, bl_start=bl_start, bl_end=bl_end,
                                 gaps=gaps, short=f["short"], long=f["long"], window_size=window_size,
                                 wps_min_length=wps_min_length, wps_max_length=wps_max_length, wps_quality=wps_quality)
        return f, w

    def feature_batch(self, items, quality_threshold=30, min_length=None, max_length=None,
                      intersect_policy="midpoint"):
        """Prepare a multi-contig window-feature batch (``ftk_window_features_batch``): ``items`` is a list of
        dicts ``name, starts, stops[, bl_start, bl_end, gaps]``.  Returns an opaque handle holding the C item
        array (and the arrays it points to) for ``window_features_batch``."""
        n = len(items)
        arr = (L.FeatureItem * n)()
        keep = []
        for i, it in enumerate(items):
            ws, we = _win(it["starts"], L.OPEN_LO), _win(it["stops"], L.OPEN_HI)
            g = L.make_gaps(it.get("gaps"))
            bs = be = None
            if it.get("bl_start") is not None and len(it["bl_start"]):
                bs = np.ascontiguousarray(it["bl_start"], dtype=np.int32)
                be = np.ascontiguousarray(it["bl_end"], dtype=np.int32)
            keep.append((ws, we, bs, be, g))
            arr[i].contig_id = self.contig_id(it["name"])
            arr[i].n_win = len(ws)
            arr[i].w_start = ws.ctypes.data
            arr[i].w_end = we.ctypes.data
            arr[i].bl_start = None if bs is None else bs.ctypes.data
            arr[i].bl_end = None if be is None else be.ctypes.data
            arr[i].n_bl = 0 if bs is None else len(bs)
            arr[i].gaps = C.pointer(g)
        f = L.make_filter(quality_threshold, min_length, max_length, intersect_policy,
                          L.FETCH_BAM_READ1 if items and self.is_bam(items[0]["name"]) else L.FETCH_TABIX)
        return dict(arr=arr, n=n, keep=keep, filter=f, rows=int(sum(len(k[0]) for k in keep)))

    def window_features_batch(self, batch, coverage=None, hist=None, hist_bins=None, overflow=None, delfi_q=30,
                              short=None, long=None):
        """Run a prepared batch; every output is a device pointer (int) or a host numpy array sized for
        ``batch["rows"]`` rows (``hist``: rows x n_bins uint32 with ``hist_bins=(len_lo, n_bins)``), or None."""
        len_lo, n_bins = hist_bins if hist_bins is not None else (0, 0)
        self._check(self.lib.ftk_window_features_batch(
            self.ctx, batch["arr"], batch["n"], C.byref(batch["filter"]), L.ptr(coverage), int(len_lo), int(n_bins),
            L.ptr(hist), L.ptr(overflow), int(delfi_q), L.ptr(short), L.ptr(long)))

    def wps_window_features(self, name: str, chrom_size: int, win_start: int, win_len: int, n_win: int,
                            window_size=120, min_length=120, max_length=180, quality_threshold=30, wps_out=None,
                            feat_quality=30, feat_min_length=None, feat_max_length=None, coverage=None, hist=None,
                            hist_bins=None, overflow=None, delfi_q=30, bl_start=None, bl_end=None, gaps=None,
                            short=None, long=None):
        """Whole-contig WPS and the window features of the regular tiling ``[win_start + k*win_len, ...)`` in ONE
        pass (``ftk_wps_window_features``).  Outputs: numpy arrays or device pointers; returns ``wps_out``
        (allocated on the host when None)."""
        if wps_out is None:
            wps_out = np.zeros(int(chrom_size), np.int64)
        f = self._filter(name, feat_quality, feat_min_length, feat_max_length, "midpoint")
        len_lo, n_bins = hist_bins if hist_bins is not None else (0, 0)
        bs = be = None
        n_bl = 0
        if bl_start is not None and len(bl_start):
            bs = np.ascontiguousarray(bl_start, dtype=np.int32)
            be = np.ascontiguousarray(bl_end, dtype=np.int32)
            n_bl = len(bs)
        g = L.make_gaps(gaps)
        self._check(self.lib.ftk_wps_window_features(
            self.ctx, self.contig_id(name), 0, int(chrom_size), int(chrom_size), int(window_size), int(min_length),
            int(max_length), int(quality_threshold), L.ptr(wps_out), int(win_start), int(win_len), int(n_win),
            C.byref(f), L.ptr(coverage), int(len_lo), int(n_bins), L.ptr(hist), L.ptr(overflow), int(delfi_q),
            L.ptr(bs), L.ptr(be), n_bl, C.byref(g), L.ptr(short), L.ptr(long)))
        return wps_out

    def wps_batch(self, names, starts, stops, chrom_sizes, out_offsets, out, window_size=120, min_length=120,
                  max_length=180, quality_threshold=30):
        """WPS of several (contig, interval) pairs in one launch into the device buffer ``out``."""
        ids = np.array([self.contig_id(n) for n in names], np.int32)
        s = np.ascontiguousarray(starts, dtype=np.int64)
        e = np.ascontiguousarray(stops, dtype=np.int64)
        cs = np.ascontiguousarray(chrom_sizes, dtype=np.int64)
        oo = np.ascontiguousarray(out_offsets, dtype=np.int64)
        self._check(self.lib.ftk_wps_batch(self.ctx, L.ptr(ids), L.ptr(s), L.ptr(e), L.ptr(cs), L.ptr(oo), len(ids),
                                           int(window_size), int(min_length), int(max_length), int(quality_threshold),
                                           L.ptr(out)))

    def frag_lengths(self, name: str, start, stop, quality_threshold=30, min_length=None, max_length=None,
                     intersect_policy="midpoint"):
        """Lengths of one window's passing fragments in file order."""
        ws = L.OPEN_LO if start is None else int(start)
        we = L.OPEN_HI if stop is None else int(stop)
        f = self._filter(name, quality_threshold, min_length, max_length, intersect_policy)
        n = C.c_int64()
        cid = self.contig_id(name)
        cap = 1 << 16
        while True:
            out = np.zeros(cap, np.int32)
            self._check(self.lib.ftk_frag_lengths(self.ctx, cid, ws, we, C.byref(f), L.ptr(out), cap, C.byref(n)))
            if n.value <= cap:
                return out[: n.value].copy()
            cap = int(n.value)

    def frag_select(self, name: str, start, stop, quality_threshold=30, min_length=None, max_length=None,
                    intersect_policy="midpoint"):
        """(start, end, mapq, strand) of one window's passing fragments in file order."""
        ws = L.OPEN_LO if start is None else int(start)
        we = L.OPEN_HI if stop is None else int(stop)
        f = self._filter(name, quality_threshold, min_length, max_length, intersect_policy)
        n = C.c_int64()
        cid = self.contig_id(name)
        cap = 1 << 16
        while True:
            s = np.zeros(cap, np.int32)
            e = np.zeros(cap, np.int32)
            q = np.zeros(cap, np.uint8)
            st = np.zeros(cap, np.uint8)
            self._check(self.lib.ftk_frag_select(self.ctx, cid, ws, we, C.byref(f), L.ptr(s), L.ptr(e), L.ptr(q),
                                                 L.ptr(st), cap, C.byref(n)))
            if n.value <= cap:
                k = n.value
                return s[:k].copy(), e[:k].copy(), q[:k].copy(), st[:k].copy()
            cap = int(n.value)

    def wps(self, name: str, start: int, stop: int, chrom_size: int, window_size=120, min_length=120,
            max_length=180, quality_threshold=30, out=None):
        """a7: WPS per base of [start, stop) (frag/_wps.py:156-188)."""
        n_pos = max(int(stop) - int(start), 0)
        if out is None:  # (every element is written by the call)
            res = self.result_array(n_pos, np.int64, pinned=not (_NARROW_WIRE and n_pos >= NARROW_WIRE_MIN))
        else:
            res = out
        self._check(self.lib.ftk_wps(self.ctx, self.contig_id(name), int(start), int(stop), int(chrom_size),
                                     int(window_size), int(min_length), int(max_length), int(quality_threshold),
                                     L.ptr(res)))
        return res

    def wps_async(self, name: str, start: int, stop: int, chrom_size: int, window_size=120, min_length=120,
                  max_length=180, quality_threshold=30):
        """``wps`` with the copy-back off the caller's path (``ftk_wps_async``): returns ``(scores, token)``
        at once; ``scores`` (page-locked) is valid after ``result_wait(token)``.  Loading and scoring the next
        contig overlaps the copy of this one; two results can be in flight."""
        n_pos = max(int(stop) - int(start), 0)
        res = self.result_array(n_pos, np.int64)
        tok = C.c_int(-1)
        self._check(self.lib.ftk_wps_async(self.ctx, self.contig_id(name), int(start), int(stop), int(chrom_size),
                                           int(window_size), int(min_length), int(max_length), int(quality_threshold),
                                           L.ptr(res), C.byref(tok)))
        if tok.value >= 0:
            # tokens count up; two results can be in flight (token & 1 is the library's buffer): an array registered
            # under an older token is complete - the library waited for that buffer's copy before reusing it
            self._pending[int(tok.value)] = res
            for old in [t for t in self._pending if t < tok.value - 1]:
                del self._pending[old]
        return res, int(tok.value)

    def result_wait(self, token: int):
        self._check(self.lib.ftk_result_wait(self.ctx, int(token)))
        self._pending.pop(int(token), None)

    def wps_intervals(self, name: str, starts, stops, chrom_size: int, window_size=120, min_length=120,
                      max_length=180, quality_threshold=30):
        """a8: WPS of many intervals of one contig in one launch; returns
        (scores, offsets) with interval i at scores[offsets[i]:offsets[i+1]]."""
        s = np.ascontiguousarray(starts, dtype=np.int64)
        e = np.ascontiguousarray(stops, dtype=np.int64)
        lens = np.maximum(e - s, 0)
        offs = np.zeros(len(s) + 1, np.int64)
        np.cumsum(lens, out=offs[1:])
        out = self.result_array(int(offs[-1]), np.int64)  # the intervals tile it: every element is written
        if len(s) and offs[-1] > 0:
            self._check(self.lib.ftk_wps_intervals(self.ctx, self.contig_id(name), L.ptr(s), L.ptr(e), len(s),
                                                   L.ptr(np.ascontiguousarray(offs[:-1])), int(chrom_size),
                                                   int(window_size), int(min_length), int(max_length),
                                                   int(quality_threshold), L.ptr(out)))
        return out, offs

    def cleavage_intervals(self, name: str, starts, stops, min_length=None, max_length=None, quality_threshold=30):
        """Cleavage proportion (percent) per base of many intervals of one contig in one launch
        (frag/_cleavage_profile.py:33-90,204-216); returns (proportions f64, offsets)."""
        s = np.ascontiguousarray(starts, dtype=np.int64)
        e = np.ascontiguousarray(stops, dtype=np.int64)
        offs = np.zeros(len(s) + 1, np.int64)
        np.cumsum(np.maximum(e - s, 0), out=offs[1:])
        out = self.result_array(int(offs[-1]), np.float64)  # the intervals tile it: every element is written
        if len(s) and offs[-1] > 0:
            self._check(self.lib.ftk_cleavage_intervals(
                self.ctx, self.contig_id(name), L.ptr(s), L.ptr(e), len(s), L.ptr(np.ascontiguousarray(offs[:-1])),
                L.LEN_OPEN if min_length is None else max(int(min_length), 0),
                L.LEN_OPEN if max_length is None else int(max_length), int(quality_threshold), L.ptr(out)))
        return out, offs

    def cleavage(self, name: str, start: int, stop: int, min_length=None, max_length=None, quality_threshold=30, out=None):
        """Cleavage proportion (percent) per base of ONE interval (``ftk_cleavage``: the tiles are numbered by the grid, no
        descriptor arrays); ``out``: a float64 host array or device tensor / address of ``stop - start`` elements."""
        n = max(int(stop) - int(start), 0)
        res = self.result_array(n, np.float64) if out is None else out
        if n:
            self._check(self.lib.ftk_cleavage(self.ctx, self.contig_id(name), int(start), int(stop),
                                              L.LEN_OPEN if min_length is None else max(int(min_length), 0),
                                              L.LEN_OPEN if max_length is None else int(max_length), int(quality_threshold),
                                              L.ptr(res)))
        return res

    # -- WPS post-processing --------------------------------------------------------
    def wps_adjust(self, scores, offsets, median_window_size=1000, mean=False, edge_sub=None, savgol_window_size=21,
                   savgol_poly_deg=2, savgol=True, out=None):
        """Running median/mean subtraction + Savitzky-Golay pass over score runs laid end to end
        (frag/_adjust_wps.py:25-50,119-140).  ``scores`` is a host float64 array or a device pointer
        (int) to one; run ``i`` is ``scores[offsets[i]:offsets[i+1]]`` and yields ``len_i - W`` values at
        ``offsets[i] - i*W`` of the result.  Raises ValueError where the reference does (window longer
        than a run, odd median window, Savitzky-Golay window longer than the filtered run)."""
        W = int(median_window_size)
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
        n_iv = len(offs) - 1
        lens = np.diff(offs)
        if W % 2 or W < 2:
            # data[W//2:-(W//2)] and the n-W running values only line up for even W (numpy broadcast error)
            raise ValueError(f"median_window_size ({W}) must be even: operands could not be broadcast together")
        if n_iv and W > int(lens.min()):
            raise ValueError(f"median_window_size ({W}) cannot be greater than the length of interval "
                             f"({int(lens.min())}).")
        sw = int(savgol_window_size) if savgol else 0
        coef = edge = None
        if sw:
            if n_iv and sw > int(lens.min()) - W:
                raise ValueError("If mode is 'interp', window_length must be less than or equal to the size of x.")
            coef, edge = savgol_operators(sw, int(savgol_poly_deg))
        total_out = int(offs[-1]) - n_iv * W if n_iv else 0
        if isinstance(scores, (int, np.integer)):
            sp = C.c_void_p(int(scores))
        else:
            scores = np.ascontiguousarray(scores, dtype=np.float64)
            sp = L.ptr(scores)
        res = np.empty(total_out, np.float64) if out is None else out
        op = C.c_void_p(int(res)) if isinstance(res, (int, np.integer)) else L.ptr(res)
        sub = None if edge_sub is None else np.ascontiguousarray(edge_sub, dtype=np.float64)
        if n_iv and total_out >= 0:
            self._check(self.lib.ftk_wps_adjust(self.ctx, sp, L.ptr(offs), n_iv, W, int(bool(mean)),
                                                None if sub is None else L.ptr(sub), sw,
                                                None if coef is None else L.ptr(coef),
                                                None if edge is None else L.ptr(edge), op))
        return res

    # -- reference images (DELFI GC on the device) ----------------------------------
    def ref_upload(self, key, image: np.ndarray, kind: int) -> int:
        """Upload a contig's reference image (uint8); returns the ref id, cached by ``key`` (2 images kept)."""
        refs = self.__dict__.setdefault("_refs", {})
        if key in refs:
            return refs[key]
        while len(refs) >= 2:  # images are up to ~250 MB each
            old_key = next(iter(refs))
            self.__dict__.setdefault("_ref_layouts", set()).discard(refs[old_key])
            self._check(self.lib.ftk_ref_release(self.ctx, refs.pop(old_key)))
        rid = self.__dict__.setdefault("_next_ref", 0)
        self._next_ref = rid + 1
        image = np.ascontiguousarray(image, dtype=np.uint8)
        self._check(self.lib.ftk_ref_upload(self.ctx, rid, L.ptr(image), len(image), int(kind)))
        refs[key] = rid
        return rid

    def ref_upload_file(self, key, path: str, offset: int, n_bytes: int, kind: int) -> int:
        """``ref_upload`` of ``n_bytes`` at ``offset`` of a file, read by the library (several read threads,
        page-locked chunks, asynchronous copies); cached by ``key`` like ``ref_upload``."""
        refs = self.__dict__.setdefault("_refs", {})
        if key in refs:
            return refs[key]
        while len(refs) >= 2:
            old_key = next(iter(refs))
            self.__dict__.setdefault("_ref_layouts", set()).discard(refs[old_key])
            self._check(self.lib.ftk_ref_release(self.ctx, refs.pop(old_key)))
        rid = self.__dict__.setdefault("_next_ref", 0)
        self._next_ref = rid + 1
        self._check(self.lib.ftk_ref_upload_file(self.ctx, rid, str(path).encode(), int(offset), int(n_bytes), int(kind)))
        refs[key] = rid
        return rid

    def ref_set_layout(self, rid: int, chrom_len: int, line_bases: int = 0, line_width: int = 0, n_starts=None,
                       n_ends=None):
        """Geometry of an uploaded image: FASTA line layout or the 2bit record's N blocks."""
        ns = np.ascontiguousarray(n_starts if n_starts is not None else [], dtype=np.int32)
        ne = np.ascontiguousarray(n_ends if n_ends is not None else [], dtype=np.int32)
        self._check(self.lib.ftk_ref_set_layout(self.ctx, rid, int(chrom_len), int(line_bases), int(line_width),
                                                L.ptr(ns) if len(ns) else None, L.ptr(ne) if len(ne) else None,
                                                len(ns)))

    def motif_counts(self, name: str, rid: int, starts, stops, k: int, fwd_offset: int, rev_offset: int,
                     both_strands: bool, negative_strand: bool, guard: int, rev_oob_is_error: bool,
                     quality_threshold: int, bam: bool = False):
        """k-mer histograms per window (frag/_end_motifs.py:118-176, frag/_breakpoint_motifs.py:124-185);
        returns (counts uint32 [n_win, 4**k], fetched fragments per window, reverse-end errors per window)."""
        ws = np.ascontiguousarray(starts, dtype=np.int32)
        we = np.ascontiguousarray(stops, dtype=np.int32)
        n = len(ws)
        counts = np.zeros((n, 4 ** int(k)), np.uint32)
        nfrag = np.zeros(n, np.int64)
        err = np.zeros(n, np.int64)
        if n:
            m = L.Motif(int(k), int(fwd_offset), int(rev_offset), int(bool(both_strands)),
                        int(bool(negative_strand)), int(guard), int(bool(rev_oob_is_error)))
            self._check(self.lib.ftk_motif_counts(self.ctx, self.contig_id(name), rid, L.ptr(ws), L.ptr(we), n,
                                                  C.byref(m), int(quality_threshold),
                                                  L.FETCH_BAM_READ1 if bam else L.FETCH_TABIX, L.ptr(counts),
                                                  L.ptr(nfrag), L.ptr(err)))
        return counts, nfrag, err

    def ref_gc_counts(self, rid: int, lo, hi):
        lo = np.ascontiguousarray(lo, dtype=np.int64)
        hi = np.ascontiguousarray(hi, dtype=np.int64)
        out = np.zeros(len(lo), np.int64)
        if len(lo):
            self._check(self.lib.ftk_ref_gc_counts(self.ctx, rid, L.ptr(lo), L.ptr(hi), len(lo), L.ptr(out)))
        return out

