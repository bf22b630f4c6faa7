"""
GPU parity: every kernel, through the C ABI, against the C oracle on the same
seeded synthetic fragments.  Integer results must be bit-exact.
"""
import os

import numpy as np
import pytest

from finaletoolkit_amd import synth
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CONTIG_LEN = 3_000_000


@pytest.fixture(scope="module")
def data(engine):
    s, e, q, st = synth.synth_contig(CONTIG_LEN, depth=30.0, seed=7)
    # sprinkle short (< 120) and very short fragments so every WPS branch is hit
    engine.load_contig("synA", s, e, q, st)
    d = dict(s=s, e=e, q=q, st=st, fr=O.Frags(s, e, q, st))
    # the same fragments as BAM contigs (read1 fetch rule, io/alignment.py:245).  synBAM: read1 spans inside their
    # fragments (forward -> at the fragment start, reverse -> at its end), what TLEN reconstruction gives for
    # ordinary pairs - the kernels then read the read1 columns only for fragments crossing a window bound.
    # synBAMx: one read1 in six sticks out of its fragment (TLEN shorter than the read's alignment), so the
    # columns are read for every fragment.
    rl = np.minimum(e - s, 100)
    r1s = np.where(st == 1, s, e - rl).astype(np.int32)
    r1e = (r1s + rl).astype(np.int32)
    engine.load_contig("synBAM", s, e, q, st, r1s, r1e)
    rng = np.random.default_rng(99)
    out = rng.random(len(s)) < 1 / 6
    shift = rng.integers(1, 400, len(s))
    x1s = np.where(out & (st == 0), np.maximum(r1s - shift, 0), r1s).astype(np.int32)
    x1e = np.where(out & (st == 1), r1e + shift, r1e).astype(np.int32)
    engine.load_contig("synBAMx", s, e, q, st, x1s, x1e)
    d["frs"] = {"synA": d["fr"], "synBAM": O.Frags(s, e, q, st, r1s, r1e), "synBAMx": O.Frags(s, e, q, st, x1s, x1e)}
    return d


KINDS = ["synA", "synBAM", "synBAMx"]


def _window_sets(rng):
    sets = {}
    sets["tile100k"] = synth.tiling_windows(CONTIG_LEN, 100_000)
    sets["tile400"] = tuple(a[:5000] for a in synth.tiling_windows(CONTIG_LEN, 400))
    ws = rng.integers(0, CONTIG_LEN - 10, 700).astype(np.int32)
    we = (ws + rng.integers(1, 250_000, 700)).astype(np.int32)
    sets["random_overlapping"] = (ws, we)
    # degenerate / malformed / open-ended windows
    ws = np.array([0, 500, 1000, 2_999_000, 100, O.OPEN_LO, 5000, O.OPEN_LO], np.int32)
    we = np.array([CONTIG_LEN, 500, 900, 4_000_000, 101, 70_000, O.OPEN_HI, O.OPEN_HI], np.int32)
    sets["edge"] = (ws, we)
    sets["one_big"] = (np.array([0], np.int32), np.array([CONTIG_LEN], np.int32))
    return sets


@pytest.mark.parametrize("policy", ["midpoint", "any"])
@pytest.mark.parametrize("flt", [dict(mapq_min=30), dict(mapq_min=0, min_len=120, max_len=180),
                                 dict(mapq_min=60, min_len=None, max_len=150), dict(mapq_min=10, min_len=300)])
def test_window_counts(engine, data, policy, flt):
    rng = np.random.default_rng(11)
    for name, (ws, we) in _window_sets(rng).items():
        want = O.c_window_counts(data["fr"], ws, we, policy=policy, **flt)
        got = engine.window_counts("synA", ws, we, quality_threshold=flt["mapq_min"],
                                   min_length=flt.get("min_len"), max_length=flt.get("max_len"),
                                   intersect_policy=policy)
        assert np.array_equal(got, want), (name, policy, flt)


def test_window_counts_total_property(engine, data):
    # tiling windows + midpoint: every passing fragment is counted exactly once
    ws, we = synth.tiling_windows(CONTIG_LEN, 100_000)
    got = engine.window_counts("synA", ws, we, quality_threshold=30)
    assert got.sum() == int((data["q"] >= 30).sum())


@pytest.mark.parametrize("kw", [dict(quality_threshold=30), dict(quality_threshold=0, min_length=120, max_length=180, intersect_policy="any"),
                                dict(quality_threshold=60, max_length=150)])
def test_fraglen_stats_on_the_device_equal_the_reference_formulas(engine, data, kw):
    """ftk_fraglen_stats (window_stats_kernel: mean / median / stdev / min / max / count / short count from the
    histogram rows on the device) against the oracle's statement of frag/_frag_length.py:156-172,202-238 on the C
    oracle's histograms: tilings, tiny windows with one, two, three fragments (the odd-count median search), overlapping
    and empty windows; on the tabix contig and on the BAM contigs (read1 fetch rule)."""
    rng = np.random.default_rng(21)
    flt = dict(mapq_min=kw.get("quality_threshold", 30), min_len=kw.get("min_length"), max_len=kw.get("max_length"),
               policy=kw.get("intersect_policy", "midpoint"))
    lo = max(kw.get("min_length") or 0, 0)
    hi = min(kw.get("max_length") or 1000, 1000)
    sets = _window_sets(rng)
    a = np.arange(1_000_000, 1_024_000, 8, dtype=np.int32)
    sets["tile8"] = (a, a + 8)  # under one fragment a window: single values, pairs, triples
    for kind in KINDS:
        for name, (ws, we) in sets.items():
            if name == "tile400":
                ws, we = ws[:1500], we[:1500]
            hist, over = O.c_fraglen_hist(data["frs"][kind], ws, we, lo, hi - lo + 1, **flt)
            assert not over.any()
            got = engine.fraglen_stats(kind, ws, we, lo, hi - lo + 1, 150, **kw)
            assert got.shape == (len(ws), 7)
            n_small = 0
            for k in range(len(ws)):
                h = hist[k]
                nz = np.nonzero(h)[0]
                if len(nz) == 0:
                    assert got[k, 5] == 0, (kind, name, k)
                    continue
                want = O.py_frag_length_stats({int(b) + lo: int(h[b]) for b in nz}, 150)
                n_small += want[5] <= 3
                assert got[k, 0] == want[0] and got[k, 1] == want[1], (kind, name, k, got[k], want)  # mean: one IEEE division of exact sums
                assert got[k, 2] == pytest.approx(want[2], rel=1e-12, abs=1e-12)                       # stdev: summation order
                assert (got[k, 3], got[k, 4], got[k, 5]) == (want[3], want[4], want[5])
                assert got[k, 6] == round(want[6] * want[5])
            if name == "tile8" and kw.get("intersect_policy", "midpoint") == "midpoint":
                assert n_small > 20  # windows with one to three fragments: the single-value and odd-count medians


@pytest.mark.parametrize("n_bins,len_lo", [(1001, 0), (64, 150), (3000, 0)])
def test_fraglen_hist(engine, data, n_bins, len_lo):
    rng = np.random.default_rng(12)
    for name, (ws, we) in _window_sets(rng).items():
        if name == "tile400":
            ws, we = ws[:800], we[:800]
        want_h, want_o = O.c_fraglen_hist(data["fr"], ws, we, len_lo, n_bins, mapq_min=30)
        got_h, got_o = engine.fraglen_hist("synA", ws, we, len_lo, n_bins, quality_threshold=30)
        assert np.array_equal(got_h, want_h), name
        assert np.array_equal(got_o, want_o), name


def test_delfi_counts(engine, data):
    rng = np.random.default_rng(13)
    ws, we = synth.tiling_windows(CONTIG_LEN, 100_000)
    we = (we - 1).astype(np.int32)  # the reference's inclusive-end bins used as exclusive stops
    bl_s = np.sort(rng.integers(0, CONTIG_LEN - 6000, 400)).astype(np.int32)
    bl_e = (bl_s + rng.integers(200, 5000, 400)).astype(np.int32)
    order = np.lexsort((bl_e, bl_s))
    bl_s, bl_e = bl_s[order], bl_e[order]
    gaps = (1_200_000, 1_500_000, [(0, 10_000), (CONTIG_LEN - 10_000, CONTIG_LEN)])
    for g in (None, gaps, (1_200_000, 1_500_000, []), (1_200_000, 1_500_000, [(0, 2_000_000)])):
        for bl in ((None, None), (bl_s, bl_e)):
            want = O.c_delfi_counts(data["fr"], ws, we, 30, bl[0], bl[1], g)
            got = engine.delfi_counts("synA", ws, we, 30, bl[0], bl[1], g)
            for a, b in zip(got, want):
                assert np.array_equal(a, b)
    # small windows (wave-per-window path) with blacklist
    ws2, we2 = synth.tiling_windows(300_000, 2_000)
    want = O.c_delfi_counts(data["fr"], ws2, we2, 20, bl_s, bl_e, gaps)
    got = engine.delfi_counts("synA", ws2, we2, 20, bl_s, bl_e, gaps)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("W,mn,mx", [(120, 120, 180), (120, 30, 400), (40, 30, 90), (121, 100, 200),
                                     (75, 20, 500), (160, 120, 150), (2, 0, 1000), (1, 0, 1000)])
def test_wps_intervals_vs_oracle(engine, data, W, mn, mx):
    rng = np.random.default_rng(W)
    ivs = [(0, 3000), (CONTIG_LEN - 2500, CONTIG_LEN), (1_000_000, 1_005_000), (4095, 4097), (8191, 12289)]
    ivs += [(int(a), int(a) + int(l)) for a, l in zip(rng.integers(0, CONTIG_LEN - 6000, 6),
                                                       rng.integers(1, 6000, 6))]
    starts = [a for a, _ in ivs]
    stops = [b for _, b in ivs]
    got, offs = engine.wps_intervals("synA", starts, stops, CONTIG_LEN, W, mn, mx, 30)
    for i, (a, b) in enumerate(ivs):
        want = O.c_wps(data["fr"], a, b, CONTIG_LEN, W, mn, mx, 30)
        assert np.array_equal(got[offs[i]:offs[i + 1]], want), (W, mn, mx, a, b)
        single = engine.wps("synA", a, b, CONTIG_LEN, W, mn, mx, 30)
        assert np.array_equal(single, want), ("single", W, mn, mx, a, b)


def test_wps_chrom_size_clip_and_degenerate(engine, data):
    # fetch window clipped by a chrom_size smaller than the data extent
    want = O.c_wps(data["fr"], 990_000, 1_000_000, 1_000_050, 120, 120, 180, 0)
    got = engine.wps("synA", 990_000, 1_000_000, 1_000_050, 120, 120, 180, 0)
    assert np.array_equal(got, want)
    assert len(engine.wps("synA", 500, 500, CONTIG_LEN)) == 0
    assert len(engine.wps("synA", 600, 500, CONTIG_LEN)) == 0


def test_wps_whole_contig_sum_property(engine, data):
    # One call over the whole contig == tiled 5 kb calls (W <= max_len: the
    # per-interval fetch cut cannot drop an influencing fragment).
    whole = engine.wps("synA", 0, 200_000, CONTIG_LEN, 120, 120, 180, 30)
    starts = list(range(0, 200_000, 5000))
    tiled, offs = engine.wps_intervals("synA", starts, [s + 5000 for s in starts], CONTIG_LEN, 120, 120, 180, 30)
    assert np.array_equal(whole, tiled)


def test_large_results_come_back_in_page_locked_memory(engine, data):
    """Results of 8 MB and more are handed out in ftk_host_alloc memory (one DMA instead of a staged copy):
    same values as a caller-provided pageable array, alive after the owner is dropped, block recycled."""
    import ctypes as C
    import gc
    from finaletoolkit_amd import engine as E
    if not E._PINNED_RESULTS:
        pytest.skip("FTK_PINNED_RESULTS=0")
    n = 1_500_000  # 12 MB of int64
    got = engine.wps("synA", 100_000, 100_000 + n, CONTIG_LEN, 120, 120, 180, 30)
    base = got
    while isinstance(base, np.ndarray) and base.base is not None:
        base = base.base
    assert isinstance(base, E._HostBlock) and got.nbytes >= E.PINNED_RESULT_MIN
    plain = np.empty(n, np.int64)
    engine.wps("synA", 100_000, 100_000 + n, CONTIG_LEN, 120, 120, 180, 30, out=plain)
    assert np.array_equal(got, plain)
    part = got[1000:2000].copy()
    view = got[1000:2000]
    ptr = base._ptr
    del got, base
    gc.collect()
    assert np.array_equal(view, part)      # the view keeps the block alive
    del view
    gc.collect()
    again = engine.result_array(n, np.int64)  # (the freed block, or another cached one of that size)
    again[:] = 7
    assert ptr and int(again.sum()) == 7 * n
    small = engine.result_array(100, np.int64)
    assert small.base is None
    del again
    gc.collect()
    # a smaller request takes the cached larger block (results shrink from contig to contig)
    smaller = engine.result_array(3 * n // 4, np.int64)  # 9 MB
    b3 = smaller
    while isinstance(b3, np.ndarray) and b3.base is not None:
        b3 = b3.base
    assert isinstance(b3, E._HostBlock)
    del smaller, b3
    gc.collect()
    # C ABI edge cases
    lib = engine.lib
    p = C.c_void_p()
    assert lib.ftk_host_alloc(-1, C.byref(p)) != 0 and lib.ftk_host_alloc(0, C.byref(p)) == 0
    lib.ftk_host_free(p)
    lib.ftk_host_free(None)
    junk = (C.c_char * 64)()
    lib.ftk_host_free(C.cast(junk, C.c_void_p))  # not one of the library's blocks: ignored


def test_wps_async_results_equal_the_synchronous_call(engine, data):
    """ftk_wps_async: kernel on the ctx stream, copy-back on the copy stream; several results in flight (the third
    call waits for the first buffer), tokens, the degenerate interval, ftk_ctx_sync as the catch-all wait."""
    spans = [(0, 1_200_000), (900_000, 2_950_000), (5, 77), (1_000_000, 2_300_000), (2_990_000, 3_000_000)]
    want = [engine.wps("synA", a, b, CONTIG_LEN, 120, 120, 180, 30) for a, b in spans]
    got = [engine.wps_async("synA", a, b, CONTIG_LEN, 120, 120, 180, 30) for a, b in spans]  # never more than 2 pending
    toks = [t for _, t in got]
    assert toks == list(range(toks[0], toks[0] + len(spans)))  # tokens count up: none names two results
    for (arr, tok), w in zip(got, want):
        engine.result_wait(tok)  # (an old token whose buffer a later call took over: complete, returns at once)
        assert np.array_equal(arr, w)
    arr, tok = engine.wps_async("synA", 500, 500, CONTIG_LEN)
    assert tok == -1 and len(arr) == 0
    engine.result_wait(tok)
    a1, _ = engine.wps_async("synA", 0, 1_500_000, CONTIG_LEN, 120, 120, 180, 30)
    a2, _ = engine.wps_async("synA", 1_500_000, 3_000_000, CONTIG_LEN, 120, 120, 180, 30)
    cov = engine.window_features("synA", np.array([0], np.int32), np.array([CONTIG_LEN], np.int32), 30)["coverage"]  # overlaps the copies
    engine.sync()
    whole = engine.wps("synA", 0, 3_000_000, CONTIG_LEN, 120, 120, 180, 30)
    assert np.array_equal(np.concatenate([a1, a2]), whole) and int(cov[0]) > 0
    with pytest.raises(Exception):
        engine.result_wait(toks[-1] + 1000)  # never handed out


def test_page_locked_result_limit_falls_back_to_ordinary_memory():
    """Beyond FTK_PINNED_RESULT_LIMIT_MB of outstanding results ftk_host_alloc refuses and result_array hands
    out an ordinary numpy array (child process: the limit is read once)."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import numpy as np\n"
            "from finaletoolkit_amd import engine as E\n"
            "eng = E.Engine(0)\n"
            "def owner(a):\n"
            "    while isinstance(a, np.ndarray) and a.base is not None: a = a.base\n"
            "    return a\n"
            "a = eng.result_array(3 << 20, np.int64)   # 24 MB: page-locked\n"
            "b = eng.result_array(3 << 20, np.int64)   # 48 MB outstanding > 40 MB: ordinary\n"
            "assert isinstance(owner(a), E._HostBlock) and not isinstance(owner(b), E._HostBlock)\n"
            "a[:] = 1; b[:] = 2\n"
            "del a\n"
            "c = eng.result_array(3 << 20, np.int64)   # room again\n"
            "assert isinstance(owner(c), E._HostBlock)\n"
            "print('ok')\n") % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                       env=dict(os.environ, FTK_PINNED_RESULT_LIMIT_MB="40"))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("policy", ["midpoint", "any"])
def test_frag_select_and_lengths(engine, data, policy):
    for (a, b) in [(100_000, 160_000), (None, 5000), (2_990_000, None), (7, 8), (None, None)]:
        ws, we, wq, wst = O.c_frag_select(data["fr"], a, b, mapq_min=20, min_len=100, max_len=400, policy=policy)
        gs, ge, gq, gst = engine.frag_select("synA", a, b, 20, 100, 400, policy)
        assert np.array_equal(gs, ws) and np.array_equal(ge, we)
        assert np.array_equal(gq, wq) and np.array_equal(gst, wst)
        gl = engine.frag_lengths("synA", a, b, 20, 100, 400, policy)
        assert np.array_equal(gl, we - ws)


@pytest.mark.parametrize("kind", ["synBAM", "synBAMx"])
def test_bam_read1_fetch_mode(engine, data, kind):
    """Read1 fetch rule (io/alignment.py:245) on every launch shape: read1 spans inside their fragments (the
    columns are read for boundary-crossing fragments only) and partly outside (read for all)."""
    fr = data["frs"][kind]
    rng = np.random.default_rng(5)
    for name, (ws, we) in _window_sets(rng).items():
        for policy in ("midpoint", "any"):
            want = O.c_window_counts(fr, ws, we, mapq_min=30, policy=policy)
            got = engine.window_counts(kind, ws, we, 30, None, None, policy)
            assert np.array_equal(got, want), (name, policy)
    # WPS: the interval's fetch window [start - max_len, stop + max_len) is where read1 must overlap
    for a, b, W, mn, mx in ((50_000, 58_000, 120, 120, 180), (0, 9_000, 120, 30, 400), (2_990_000, CONTIG_LEN, 60, 30, 200),
                            (1_234_567, 1_250_001, 121, 100, 300)):
        want = O.c_wps(fr, a, b, CONTIG_LEN, W, mn, mx, 30)
        assert np.array_equal(engine.wps(kind, a, b, CONTIG_LEN, W, mn, mx, 30), want), (a, b, W)
    starts = np.arange(100_000, 2_900_000, 70_001, dtype=np.int64)
    stops = starts + rng.integers(1, 9_000, len(starts))
    got, offs = engine.wps_intervals(kind, starts, stops, CONTIG_LEN, 120, 120, 180, 30)
    for i, (a, b) in enumerate(zip(starts, stops)):
        assert np.array_equal(got[offs[i]:offs[i + 1]], O.c_wps(fr, int(a), int(b), CONTIG_LEN, 120, 120, 180, 30)), i
    for a, b in ((300_000, 304_000), (0, 5_000), (2_995_000, CONTIG_LEN)):
        assert np.array_equal(engine.cleavage(kind, a, b, None, None, 20), O.c_cleavage(fr, a, b, None, None, 20)[2])
    ws, we = synth.tiling_windows(CONTIG_LEN, 100_000)
    want = O.c_delfi_counts(fr, ws, we, 30, None, None, None)
    got = engine.delfi_counts(kind, ws, we, 30)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)


def test_empty_and_errors(engine):
    from finaletoolkit_amd._lib import FtkError
    z32 = np.zeros(0, np.int32)
    z8 = np.zeros(0, np.uint8)
    engine.load_contig("empty", z32, z32, z8, z8)
    assert engine.window_counts("empty", [0, None], [100, None]).tolist() == [0, 0]
    assert engine.wps("empty", 0, 300, 1000).tolist() == [0] * 300
    h, o = engine.fraglen_hist("empty", [0], [10], 0, 10)
    assert h.sum() == 0 and o.sum() == 0
    with pytest.raises(FtkError) as ei:
        engine.load_contig("bad", np.array([5, 3], np.int32), np.array([9, 9], np.int32),
                           np.array([1, 1], np.uint8), np.array([1, 1], np.uint8))
    assert ei.value.code == -8
    with pytest.raises(FtkError):
        engine.load_contig("bad2", np.array([5], np.int32), np.array([3], np.int32), np.array([1], np.uint8))
    with pytest.raises(KeyError):
        engine.window_counts("nope", [0], [1])


def test_fused_window_features_equal_separate_calls(engine, data):
    rng = np.random.default_rng(21)
    bl_s = np.sort(rng.integers(0, CONTIG_LEN - 6000, 300)).astype(np.int32)
    bl_e = (bl_s + rng.integers(200, 5000, 300)).astype(np.int32)
    o = np.lexsort((bl_e, bl_s))
    bl_s, bl_e = bl_s[o], bl_e[o]
    gaps = (1_200_000, 1_500_000, [(0, 10_000), (CONTIG_LEN - 10_000, CONTIG_LEN)])
    for name, (ws, we) in _window_sets(rng).items():
        if name == "edge":
            continue  # open-ended windows are a coverage notion; DELFI bins are closed
        if name == "tile400":
            ws, we = ws[:600], we[:600]
        for n_bins in (301, 2500):  # per-wave LDS histograms / block histograms only
            r = engine.window_features("synA", ws, we, quality_threshold=20, min_length=50, max_length=400,
                                       hist=(50, n_bins), delfi=dict(quality_threshold=30, bl_start=bl_s, bl_end=bl_e,
                                                                     gaps=gaps))
            assert np.array_equal(r["coverage"], O.c_window_counts(data["fr"], ws, we, mapq_min=20, min_len=50,
                                                                   max_len=400)), name
            wh, wo = O.c_fraglen_hist(data["fr"], ws, we, 50, n_bins, mapq_min=20, min_len=50, max_len=400)
            assert np.array_equal(r["hist"], wh) and np.array_equal(r["overflow"], wo), name
            sh, lg, _ = O.c_delfi_counts(data["fr"], ws, we, 30, bl_s, bl_e, gaps)
            assert np.array_equal(r["short"], sh) and np.array_equal(r["long"], lg), name
    # feature subsets
    ws, we = synth.tiling_windows(CONTIG_LEN, 100_000)
    r = engine.window_features("synA", ws, we, coverage=False, delfi=dict(quality_threshold=30))
    sh, lg, _ = O.c_delfi_counts(data["fr"], ws, we, 30)
    assert set(r) == {"short", "long"} and np.array_equal(r["short"], sh) and np.array_equal(r["long"], lg)


def test_many_windows_multiblock_planner(engine, data):
    # > 16384 windows: bounds_kernel + scan_kernel path instead of the fused single-block planner
    rng = np.random.default_rng(33)
    n = 40_000
    ws = rng.integers(0, CONTIG_LEN - 10, n).astype(np.int32)
    we = (ws + rng.integers(1, 3000, n)).astype(np.int32)
    we[::1000] = ws[::1000] + 200_000  # a few large windows among the small ones
    want = O.c_window_counts(data["fr"], ws, we, mapq_min=30)
    assert np.array_equal(engine.window_counts("synA", ws, we, 30), want)
    h, o = engine.fraglen_hist("synA", ws[:20_000], we[:20_000], 100, 200, 30)
    wh, wo = O.c_fraglen_hist(data["fr"], ws[:20_000], we[:20_000], 100, 200, mapq_min=30)
    assert np.array_equal(h, wh) and np.array_equal(o, wo)


def test_contig_ids_are_not_reused_after_release(engine):
    # regression: releasing a contig must not let a later upload alias a live contig's id
    one = lambda v: (np.array([v], np.int32), np.array([v + 100], np.int32), np.array([60], np.uint8),
                     np.array([1], np.uint8))
    for k, v in (("idA", 1000), ("idB", 2000), ("idC", 3000)):
        engine.load_contig(k, *one(v))
    engine.release("idA")
    engine.load_contig("idD", *one(4000))
    for k, v in (("idB", 2000), ("idC", 3000), ("idD", 4000)):
        assert engine.frag_select(k, None, None, 0)[0].tolist() == [v]
    for k in ("idB", "idC", "idD"):
        engine.release(k)


@pytest.mark.parametrize("world", [2, 3, 7])
def test_split_units_reproduce_the_whole_contig(engine, data, world):
    """bench.py's multi-GPU split: a unit loaded with its halo of fragments gives exactly the
    whole contig's windows / bases of its range (features under the midpoint policy, DELFI, WPS)."""
    from finaletoolkit_amd.sharding import split_units, unit_halo
    ws, we = synth.tiling_windows(CONTIG_LEN, 100_000)
    rng = np.random.default_rng(3)
    bl_s = np.sort(rng.integers(0, CONTIG_LEN - 5000, 80)).astype(np.int32)
    bl_e = (bl_s + rng.integers(100, 4000, 80)).astype(np.int32)
    gaps = (1_200_000, 1_500_000, [(0, 10_000), (CONTIG_LEN - 10_000, CONTIG_LEN)])
    whole = engine.window_features("synA", ws, we, 30, hist=(0, 1001), delfi=dict(bl_start=bl_s, bl_end=bl_e, gaps=gaps))
    whole_wps = engine.wps("synA", 0, CONTIG_LEN, CONTIG_LEN)
    halo = unit_halo(int((data["e"] - data["s"]).max()), 120)
    parts = {k: [] for k in whole}
    wps_parts = []
    for r, c, a, b in split_units({"synA": CONTIG_LEN}, world, 100_000):
        lo, hi = np.searchsorted(data["s"], [a - halo, b + halo])
        name = f"unit{r}"
        engine.load_contig(name, data["s"][lo:hi], data["e"][lo:hi], data["q"][lo:hi], data["st"][lo:hi])
        m = (ws >= a) & (ws < b)
        got = engine.window_features(name, ws[m], we[m], 30, hist=(0, 1001),
                                     delfi=dict(bl_start=bl_s, bl_end=bl_e, gaps=gaps))
        for k in whole:
            parts[k].append(got[k])
        wps_parts.append(engine.wps(name, a, b, CONTIG_LEN))
        engine.release(name)
    for k in whole:
        assert np.array_equal(np.concatenate(parts[k]), whole[k]), k
    assert np.array_equal(np.concatenate(wps_parts), whole_wps)


def test_batched_launches_equal_per_contig_calls(engine, data):
    """ftk_window_features_batch / ftk_wps_batch over several contigs == the per-contig calls."""
    rng = np.random.default_rng(11)
    names, sizes = ["synA"], [CONTIG_LEN]
    for k, size in enumerate((700_000, 1_250_000, 90_000)):
        s, e, q, st = synth.synth_contig(size, depth=20.0 + 5 * k, seed=20 + k)
        engine.load_contig(f"bat{k}", s, e, q, st)
        names.append(f"bat{k}")
        sizes.append(size)
    items, want = [], []
    for name, size in zip(names, sizes):
        ws, we = synth.tiling_windows(size, 50_000)
        bl_s = np.sort(rng.integers(0, size - 5000, 30)).astype(np.int32)
        bl_e = (bl_s + rng.integers(100, 4000, 30)).astype(np.int32)
        gaps = (size // 3, size // 3 + 40_000, [(0, 5_000), (size - 5_000, size)])
        items.append(dict(name=name, starts=ws, stops=we, bl_start=bl_s, bl_end=bl_e, gaps=gaps))
        want.append(engine.window_features(name, ws, we, 30, hist=(0, 601),
                                           delfi=dict(bl_start=bl_s, bl_end=bl_e, gaps=gaps)))
    batch = engine.feature_batch(items, quality_threshold=30)
    rows = batch["rows"]
    cov, over = np.zeros(rows, np.int64), np.zeros(rows, np.int64)
    hist = np.zeros((rows, 601), np.uint32)
    sh, lg = np.zeros(rows, np.int64), np.zeros(rows, np.int64)
    for _ in range(2):  # second call re-uses the cached device descriptors
        engine.window_features_batch(batch, coverage=cov, hist=hist, hist_bins=(0, 601), overflow=over, short=sh, long=lg)
        for key, got in (("coverage", cov), ("hist", hist), ("overflow", over), ("short", sh), ("long", lg)):
            assert np.array_equal(got, np.concatenate([w[key] for w in want])), key
    cov2 = np.zeros(rows, np.int64)  # coverage only
    engine.window_features_batch(batch, coverage=cov2)
    assert np.array_equal(cov2, cov)
    # WPS: whole contigs plus a sub-range, one launch
    iv = [(n, 0, s, s) for n, s in zip(names, sizes)] + [("synA", 1_000_123, 1_004_500, CONTIG_LEN)]
    offs = np.concatenate([[0], np.cumsum([b - a for _, a, b, _ in iv])]).astype(np.int64)
    got = np.zeros(int(offs[-1]), np.int64)
    engine.wps_batch([x[0] for x in iv], [x[1] for x in iv], [x[2] for x in iv], [x[3] for x in iv], offs[:-1], got)
    for k, (n, a, b, cs) in enumerate(iv):
        assert np.array_equal(got[offs[k]:offs[k + 1]], engine.wps(n, a, b, cs)), n
    for n in names[1:]:
        engine.release(n)


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_feature_fuzz_extreme_parameters(engine, data, seed):
    """Random filter / window / histogram parameters drawn from extreme values (open bounds, bounds beyond
    2^30, negative and > 255 mapq cuts, inverted and negative windows) against the C oracle."""
    rng = np.random.default_rng(1000 + seed)
    ext = [O.OPEN_LO, -(2 ** 30) - 5, -1, 0, 1, 5000, CONTIG_LEN - 1, CONTIG_LEN, CONTIG_LEN + 7, 2 ** 30 - 1, 2 ** 30,
           2 ** 30 + 9, O.OPEN_HI]
    for it in range(6):
        n = int(rng.integers(1, 400))
        ws = rng.integers(-50_000, CONTIG_LEN + 50_000, n).astype(np.int64)
        we = ws + rng.integers(-2000, 300_000, n)
        k = rng.integers(0, n, max(1, n // 5))
        ws[k] = rng.choice(ext, len(k))
        k = rng.integers(0, n, max(1, n // 5))
        we[k] = rng.choice(ext, len(k))
        ws = np.clip(ws, O.OPEN_LO, O.OPEN_HI).astype(np.int32)
        we = np.clip(we, O.OPEN_LO, O.OPEN_HI).astype(np.int32)
        if it == 0:  # a bin tiling large enough for the block-per-window path, with a few extremes mixed in
            ws, we = synth.tiling_windows(CONTIG_LEN, 5_000)
            ws, we = ws.copy(), we.copy()
            ws[::97] = O.OPEN_LO
            we[::89] = O.OPEN_HI
        mapq = int(rng.choice([-5, 0, 1, 30, 59, 60, 61, 255, 256, 1000]))
        mn = rng.choice([None, 0, 1, 100, 167, 1000, 2 ** 30, 2 ** 31 - 1])
        mx = rng.choice([None, 0, 150, 167, 999, 1000, 2 ** 30, 2 ** 31 - 1])
        mn = None if mn is None else int(mn)
        mx = None if mx is None else int(mx)
        policy = str(rng.choice(["midpoint", "any"]))
        len_lo = int(rng.choice([-50, 0, 100, 167, 990, 5000]))
        n_bins = int(rng.choice([1, 7, 64, 1001, 2500]))
        flt = dict(mapq_min=mapq, min_len=mn, max_len=mx, policy=policy)
        want_c = O.c_window_counts(data["fr"], ws, we, **flt)
        want_h, want_o = O.c_fraglen_hist(data["fr"], ws, we, len_lo, n_bins, **flt)
        got = engine.window_features("synA", ws, we, mapq, mn, mx, policy, hist=(len_lo, n_bins))
        tag = (seed, it, flt, len_lo, n_bins)
        assert np.array_equal(got["coverage"], want_c), tag
        assert np.array_equal(got["hist"], want_h), tag
        assert np.array_equal(got["overflow"], want_o), tag
        dq = int(rng.choice([-1, 0, 30, 300]))
        gaps = rng.choice([0, 1, 2])
        g = None if gaps == 0 else ((int(rng.integers(0, CONTIG_LEN)), int(rng.integers(0, CONTIG_LEN)), [])
                                    if gaps == 1 else (1_000_000, 1_400_000, [(0, 20_000), (CONTIG_LEN - 30_000, 2 ** 30)]))
        want = O.c_delfi_counts(data["fr"], ws, we, dq, None, None, g)
        got = engine.delfi_counts("synA", ws, we, dq, None, None, g)
        for a, b in zip(got, want):
            assert np.array_equal(a, b), (tag, dq, g)


# FTK_FEAT_BLOCK=1 makes the library take the block-per-window kernels for ANY window set (the host otherwise
# reserves them for many windows of similar length); test_block_kernels_forced_* re-runs the feature tests
# of this module that way, with the FAST kernels on and off.
FORCED_BLOCK_PATH = os.environ.get("FTK_FEAT_BLOCK") == "1"


@pytest.mark.parametrize("fast", ["1", "0"])
def test_block_kernels_forced_on_every_window_set(fast):
    if FORCED_BLOCK_PATH:
        pytest.skip("this is the forced run")
    import subprocess
    import sys
    env = dict(os.environ, FTK_FEAT_BLOCK="1", FTK_FEAT_FAST=fast)
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", os.path.abspath(__file__), "-k",
                        "fast_block_kernels or feature_fuzz or window_counts or fraglen_hist or delfi_counts or upper_limit"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_fast_block_kernels_against_the_oracle(engine, data, seed, kind):
    """The FAST block-per-window kernels (midpoint policy, no length bounds, one mapq cut: the request of the
    whole-genome pass) on tilings of several bin widths with extreme and inverted windows mixed in; coverage,
    histogram and DELFI separately and fused, blacklist CSR and gap intervals near and far from the windows.
    On a tabix-style contig and on BAM contigs (read1 fetch rule) with read1 spans inside / partly outside
    their fragments."""
    rng = np.random.default_rng(3000 + seed)
    data = dict(data, fr=data["frs"][kind])
    for width in (100_000, 5_000, 1_237, 20_000):
        ws, we = synth.tiling_windows(CONTIG_LEN, width)
        ws, we = ws.copy(), we.copy()
        if len(ws) < 300:  # the block path wants at least one window per CU: repeat the tiling, shifted
            reps = -(-300 // len(ws))
            ws = np.concatenate([np.clip(ws + 977 * r, 0, CONTIG_LEN) for r in range(reps)]).astype(np.int32)
            we = np.concatenate([np.clip(we + 977 * r, 0, CONTIG_LEN) for r in range(reps)]).astype(np.int32)
        k = rng.integers(0, len(ws), 12)
        ws[k[8]], we[k[8]] = 500_000, 400_000          # inverted
        ws[k[9]], we[k[9]] = 123_456, 123_456          # empty
        ws[k[10]], we[k[10]] = -7, width // 2          # hangs over the contig start
        ws[k[11]], we[k[11]] = 2 ** 30, 2 ** 30 + 99   # beyond every coordinate
        if FORCED_BLOCK_PATH:  # open and huge windows: the host's shape heuristic would send these to the planner
            ws[k[:4]] = [O.OPEN_LO, -7, 0, 2 ** 30]
            we[k[4:8]] = [O.OPEN_HI, 2 ** 30 + 3, 0, -5]
        q = int(rng.choice([0, 1, 30, 60, 61, 255, 300, -3]))
        len_lo = int(rng.choice([0, 0, 100, 167, -20]))
        n_bins = int(rng.choice([1001, 1001, 64, 1, 2047]))
        tag = (seed, width, q, len_lo, n_bins)
        want_c = O.c_window_counts(data["fr"], ws, we, mapq_min=q)
        want_h, want_o = O.c_fraglen_hist(data["fr"], ws, we, len_lo, n_bins, mapq_min=q)
        assert np.array_equal(engine.window_counts(kind, ws, we, q), want_c), tag
        gh, go = engine.fraglen_hist(kind, ws, we, len_lo, n_bins, q)
        assert np.array_equal(gh, want_h) and np.array_equal(go, want_o), tag
        bl_s = np.sort(rng.integers(0, CONTIG_LEN - 6000, 300)).astype(np.int32)
        bl_e = (bl_s + rng.integers(1, 5000, 300)).astype(np.int32)
        order = np.lexsort((bl_e, bl_s))
        bl_s, bl_e = bl_s[order], bl_e[order]
        for gaps in (None, (1_000_050, 1_400_020, [(0, 20_010), (CONTIG_LEN - 30_000, 2 ** 30)]),
                     (int(rng.integers(0, CONTIG_LEN)), int(rng.integers(0, CONTIG_LEN)), []),
                     (700_000, 650_000, [(5, 2_000_000), (1_900_000, 2_100_000)])):
            for bl in ((None, None), (bl_s, bl_e)):
                want = O.c_delfi_counts(data["fr"], ws, we, q, bl[0], bl[1], gaps)
                got = engine.delfi_counts(kind, ws, we, q, bl[0], bl[1], gaps)
                for a, b in zip(got, want):
                    assert np.array_equal(a, b), (tag, gaps, bl[0] is None)
            fused = engine.window_features(kind, ws, we, q, hist=(len_lo, n_bins),
                                           delfi=dict(quality_threshold=q, bl_start=bl_s, bl_end=bl_e, gaps=gaps))
            want = O.c_delfi_counts(data["fr"], ws, we, q, bl_s, bl_e, gaps)
            assert np.array_equal(fused["coverage"], want_c) and np.array_equal(fused["hist"], want_h), (tag, gaps)
            assert np.array_equal(fused["overflow"], want_o), (tag, gaps)
            assert np.array_equal(fused["short"], want[0]) and np.array_equal(fused["long"], want[1]), (tag, gaps)
        # a different DELFI mapq cut takes the general kernels: same answers
        fused = engine.window_features(kind, ws, we, q, hist=(len_lo, n_bins), delfi=dict(quality_threshold=17))
        want = O.c_delfi_counts(data["fr"], ws, we, 17)
        assert np.array_equal(fused["coverage"], want_c) and np.array_equal(fused["short"], want[0]), tag
        assert np.array_equal(fused["long"], want[1]), tag


def test_fast_block_kernels_near_the_upper_limit(engine):
    """Fragments just below 2^30 on a tiling wide enough for the block path: the doubled midpoint test must
    not overflow (m2 = fs + fe reaches 2^31 - 2)."""
    rng = np.random.default_rng(6)
    top = 2 ** 30 - 1
    n = 60_000
    s = np.sort(rng.integers(top - 3_000_000, top - 600, n)).astype(np.int32)
    e = np.minimum(s + rng.integers(0, 600, n), top).astype(np.int32)
    q = rng.integers(0, 61, n).astype(np.uint8)
    s[-1], e[-1] = top, top
    engine.load_contig("hi2", s, e, q, np.zeros(n, np.uint8))
    fr = O.Frags(s, e, q, np.zeros(n, np.uint8))
    ws = np.arange(top - 3_000_000, top + 1, 9_973, dtype=np.int64)
    we = np.minimum(ws + 9_973, 2 ** 30 + 5)
    ws, we = ws.astype(np.int32), we.astype(np.int32)
    ws[0], we[-1] = O.OPEN_LO, O.OPEN_HI
    assert len(ws) >= 300
    assert np.array_equal(engine.window_counts("hi2", ws, we, 10), O.c_window_counts(fr, ws, we, mapq_min=10))
    h, o = engine.fraglen_hist("hi2", ws, we, 0, 700, 0)
    wh, wo = O.c_fraglen_hist(fr, ws, we, 0, 700, mapq_min=0)
    assert np.array_equal(h, wh) and np.array_equal(o, wo)
    g = (top - 1_000_000, top - 900_000, [(top - 50_000, top)])
    for a, b in zip(engine.delfi_counts("hi2", ws, we, 0, None, None, g), O.c_delfi_counts(fr, ws, we, 0, None, None, g)):
        assert np.array_equal(a, b)
    engine.release("hi2")


def test_coordinates_near_the_upper_limit(engine):
    """Fragments and windows just below 2^30 (the largest coordinate the SoA admits)."""
    rng = np.random.default_rng(5)
    top = 2 ** 30 - 1
    n = 20_000
    s = np.sort(rng.integers(top - 3_000_000, top - 600, n)).astype(np.int32)
    e = np.minimum(s + rng.integers(0, 600, n), top).astype(np.int32)
    q = rng.integers(0, 61, n).astype(np.uint8)
    st = rng.integers(0, 2, n).astype(np.uint8)
    s[-1], e[-1] = top, top  # zero-length fragment at the very end
    engine.load_contig("hi", s, e, q, st)
    fr = O.Frags(s, e, q, st)
    ws = np.array([top - 3_000_000, top - 100_000, top - 1, top, O.OPEN_LO, top - 2_000_000], np.int32)
    we = np.array([top - 2_000_000, top, top, top + 1, O.OPEN_HI, 2 ** 30 + 5], np.int32)
    for policy in ("midpoint", "any"):
        assert np.array_equal(engine.window_counts("hi", ws, we, 10, intersect_policy=policy),
                              O.c_window_counts(fr, ws, we, mapq_min=10, policy=policy)), policy
    h, o = engine.fraglen_hist("hi", ws, we, 0, 700, quality_threshold=0)
    wh, wo = O.c_fraglen_hist(fr, ws, we, 0, 700, mapq_min=0)
    assert np.array_equal(h, wh) and np.array_equal(o, wo)
    g = (top - 1_000_000, top - 900_000, [(top - 50_000, top)])
    for a, b in zip(engine.delfi_counts("hi", ws, we, 0, None, None, g), O.c_delfi_counts(fr, ws, we, 0, None, None, g)):
        assert np.array_equal(a, b)
    a, b = top - 20_000, top
    assert np.array_equal(engine.wps("hi", a, b, 2 ** 30), O.c_wps(fr, a, b, 2 ** 30, 120, 120, 180, 30))
    engine.release("hi")


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_wps_and_cleavage_fuzz(engine, data, seed):
    """Random WPS / cleavage parameters incl. degenerate ones (window 1..2000, min > max, chrom_size inside the
    data, intervals hanging over both contig ends) against the C oracle."""
    rng = np.random.default_rng(2000 + seed)
    for it in range(10):
        W = int(rng.choice([1, 2, 3, 60, 119, 120, 121, 500, 2000]))
        mn = int(rng.choice([0, 1, 100, 120, 167, 400]))
        mx = int(rng.choice([0, 90, 150, 167, 180, 600, 5000]))
        q = int(rng.choice([0, 1, 30, 60, 61]))
        chrom = int(rng.choice([CONTIG_LEN, CONTIG_LEN - 123_456, 1_000_000, CONTIG_LEN + 5000]))
        a = int(rng.choice([0, -300, 5, 999_000, CONTIG_LEN - 2000, int(rng.integers(0, CONTIG_LEN))]))
        b = a + int(rng.choice([1, 2, 4095, 4096, 4097, 9000, 20_000]))
        want = O.c_wps(data["fr"], a, b, chrom, W, mn, mx, q)
        got = engine.wps("synA", a, b, chrom, W, mn, mx, q)
        assert np.array_equal(got, want), ("wps", seed, it, W, mn, mx, q, chrom, a, b)
        a2 = max(a, 0)
        b2 = a2 + (b - a)
        lo = rng.choice([None, 0, 100, 167])
        hi = rng.choice([None, 150, 167, 600])
        lo = None if lo is None else int(lo)
        hi = None if hi is None else int(hi)
        want = O.c_cleavage(data["fr"], a2, b2, lo, hi, q)[2]  # (depth, ends, proportion)
        got = engine.cleavage("synA", a2, b2, lo, hi, q)
        assert np.array_equal(got, want), ("cleavage", seed, it, lo, hi, q, a2, b2)


def test_cleavage_of_a_very_deep_region(engine):
    """chrM-like depth: 120 000 fragments over 12 kb (> 32 768 candidates per 4 096-base tile, where the kernel's packed
    16-bit LDS counters would not be exact: those tiles take the 32-bit half-tile path), beside an ordinary region of
    the same contig (16-bit path), a pile of 70 000 fragment ends on ONE base, and intervals that start inside a tile."""
    rng = np.random.default_rng(77)
    deep_s = rng.integers(20_000, 32_000, 120_000)
    pile_s = np.full(70_000, 50_000)
    calm_s = rng.integers(60_000, 400_000, 60_000)
    s = np.concatenate([deep_s, pile_s, calm_s])
    e = s + np.concatenate([rng.integers(60, 400, len(deep_s)), rng.integers(100, 300, len(pile_s)), rng.integers(60, 400, len(calm_s))])
    order = np.argsort(s, kind="stable")
    s, e = s[order].astype(np.int32), e[order].astype(np.int32)
    q = rng.integers(0, 61, len(s)).astype(np.uint8)
    st = rng.integers(0, 2, len(s)).astype(np.uint8)
    st[(s == 50_000)] = 1  # the pile: + strand, so 70 000 ends land on base 50 000
    engine.load_contig("deep", s, e, q, st)
    fr = O.Frags(s, e, q, st)
    for a, b, lo, hi, mq in ((0, 420_000, None, None, 0), (19_000, 36_000, 100, 300, 20), (49_000, 53_000, None, None, 0),
                             (25_000, 25_001, None, None, 0), (23_456, 31_111, 0, 167, 30)):
        want = O.c_cleavage(fr, a, b, lo, hi, mq)[2]
        got = engine.cleavage("deep", a, b, lo, hi, mq)
        assert np.array_equal(got, want), (a, b, lo, hi, mq)
    engine.release("deep")


def test_one_call_file_loaders(engine, tmp_path):
    """ftk_frags_load_fraggz / ftk_frags_load_bam: file -> HBM in one C call, contig ids in file order."""
    import ctypes as C
    import os
    from finaletoolkit_amd import _lib as L
    from tests.helpers import DATA, GOLDEN, read_frag_gz, write_synthetic_bam
    lib = engine.lib
    n = C.c_int()
    base = 7000
    assert lib.ftk_frags_load_fraggz(engine.ctx, os.path.join(GOLDEN, "synth.frag.gz").encode(), None, 4, base,
                                     C.byref(n)) == 0
    want = read_frag_gz(os.path.join(GOLDEN, "synth.frag.gz"))
    assert n.value == len(want)
    for k in range(n.value):
        name = lib.ftk_frags_name(engine.ctx, base + k).decode()
        rows = C.c_int64()
        assert lib.ftk_frags_info(engine.ctx, base + k, C.byref(rows), None, None) == 0
        assert rows.value == len(want[name][0])
        f = L.make_filter(0, None, None, "any")
        ws, we, out = np.array([O.OPEN_LO], np.int32), np.array([O.OPEN_HI], np.int32), np.zeros(1, np.int64)
        assert lib.ftk_window_counts(engine.ctx, base + k, L.ptr(ws), L.ptr(we), 1, C.byref(f), L.ptr(out)) == 0
        assert out[0] == rows.value
        assert lib.ftk_frags_release(engine.ctx, base + k) == 0
    assert lib.ftk_frags_name(engine.ctx, base + 99) is None
    assert lib.ftk_frags_load_bam(engine.ctx, os.path.join(DATA, "12.3444.b37.bam").encode(), None, 2, base,
                                  C.byref(n)) == 0
    assert n.value == 1 and lib.ftk_frags_name(engine.ctx, base).decode() == "12"
    rows = C.c_int64()
    assert lib.ftk_frags_info(engine.ctx, base, C.byref(rows), None, None) == 0 and rows.value == 17
    assert lib.ftk_frags_release(engine.ctx, base) == 0
    assert lib.ftk_frags_load_fraggz(engine.ctx, str(tmp_path / "absent.gz").encode(), None, 2, base, C.byref(n)) == L.FTK_ERR_IO


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("win_len", [100_000, 5_200, 250_001])
def test_fused_wps_and_window_features_equal_separate_calls(engine, data, win_len, kind):
    """ftk_wps_window_features (one pass) == ftk_wps + ftk_window_features, for bin lengths from just above
    the minimum (tile + longest fragment) to odd sizes, with blacklist and gaps."""
    rng = np.random.default_rng(win_len)
    n_win = -(-CONTIG_LEN // win_len)
    ws = (np.arange(n_win, dtype=np.int64) * win_len).astype(np.int32)
    we = (ws.astype(np.int64) + win_len).astype(np.int32)
    bl_s = np.sort(rng.integers(0, CONTIG_LEN - 5000, 120)).astype(np.int32)
    bl_e = (bl_s + rng.integers(100, 4000, 120)).astype(np.int32)
    gaps = (1_200_000, 1_500_000, [(0, 10_000), (CONTIG_LEN - 10_000, CONTIG_LEN)])
    want = engine.window_features(kind, ws, we, 25, 50, 700, hist=(20, 640),
                                  delfi=dict(quality_threshold=30, bl_start=bl_s, bl_end=bl_e, gaps=gaps))
    want_wps = engine.wps(kind, 0, CONTIG_LEN, CONTIG_LEN)
    cov, over = np.zeros(n_win, np.int64), np.zeros(n_win, np.int64)
    hist = np.zeros((n_win, 640), np.uint32)
    sh, lg = np.zeros(n_win, np.int64), np.zeros(n_win, np.int64)
    got_wps = engine.wps_window_features(kind, CONTIG_LEN, 0, win_len, n_win, feat_quality=25, feat_min_length=50,
                                         feat_max_length=700, coverage=cov, hist=hist, hist_bins=(20, 640),
                                         overflow=over, delfi_q=30, bl_start=bl_s, bl_end=bl_e, gaps=gaps, short=sh,
                                         long=lg)
    assert np.array_equal(got_wps, want_wps)
    for key, got in (("coverage", cov), ("hist", hist), ("overflow", over), ("short", sh), ("long", lg)):
        assert np.array_equal(got, want[key]), (key, win_len)
    # coverage only / DELFI only
    cov2 = np.zeros(n_win, np.int64)
    engine.wps_window_features(kind, CONTIG_LEN, 0, win_len, n_win, feat_quality=25, feat_min_length=50,
                               feat_max_length=700, coverage=cov2)
    assert np.array_equal(cov2, cov)
    sh2, lg2 = np.zeros(n_win, np.int64), np.zeros(n_win, np.int64)
    engine.wps_window_features(kind, CONTIG_LEN, 0, win_len, n_win, delfi_q=30, bl_start=bl_s, bl_end=bl_e,
                               gaps=gaps, short=sh2, long=lg2)
    assert np.array_equal(sh2, sh) and np.array_equal(lg2, lg)
    # bins that start inside the contig and stop before its end: the rest of the contig is ignored
    sub = np.zeros(5, np.int64)
    engine.wps_window_features(kind, CONTIG_LEN, win_len, win_len, 5, feat_quality=25, feat_min_length=50,
                               feat_max_length=700, coverage=sub)
    assert np.array_equal(sub, cov[1:6])
    from finaletoolkit_amd import _lib as L
    with pytest.raises(L.FtkError):  # bins shorter than tile + longest fragment
        engine.wps_window_features(kind, CONTIG_LEN, 0, 4_000, 10, coverage=np.zeros(10, np.int64))


@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("win_len", [100_000, 37_777])
def test_features_and_wps_in_one_launch_equal_the_two_calls(engine, data, win_len, kind):
    """ftk_window_features_wps: the merged launch (feature blocks first, WPS tiles behind them in the same grid)
    gives exactly the results of ftk_window_features followed by ftk_wps -- coverage + histogram + DELFI with
    blacklist and gaps, coverage alone, DELFI alone; a request the FAST block path does not serve (length bounds
    on the coverage filter, `any` policy) and a host WPS array fall back to the two launches with the same results;
    an interval that is a sub-range of the contig; odd W."""
    import torch
    rng = np.random.default_rng(win_len)
    ws, we = synth.tiling_windows(CONTIG_LEN, win_len)
    n_win = len(ws)
    bl_s = np.sort(rng.integers(0, CONTIG_LEN - 5000, 150)).astype(np.int32)
    bl_e = (bl_s + rng.integers(100, 4000, 150)).astype(np.int32)
    gaps = (1_200_000, 1_500_000, [(0, 10_000), (CONTIG_LEN - 10_000, CONTIG_LEN)])
    dev = torch.device("cuda", 0)

    def outs():
        return dict(coverage=torch.full((n_win,), -7, dtype=torch.int64, device=dev),
                    hist=torch.full((n_win, 640), 9, dtype=torch.int32, device=dev),
                    overflow=torch.full((n_win,), -7, dtype=torch.int64, device=dev),
                    short=torch.full((n_win,), -7, dtype=torch.int64, device=dev),
                    long=torch.full((n_win,), -7, dtype=torch.int64, device=dev))

    for kw, a, b, W in ((dict(quality_threshold=30), 0, CONTIG_LEN, 120),
                        (dict(quality_threshold=30), 123_457, 2_000_001, 121),
                        (dict(quality_threshold=25, min_length=50, max_length=700), 0, CONTIG_LEN, 120),  # general kernels
                        (dict(quality_threshold=30, intersect_policy="any"), 5_000, 900_000, 60)):
        want = engine.window_features(kind, ws, we, hist=(20, 640),
                                      delfi=dict(quality_threshold=30, bl_start=bl_s, bl_end=bl_e, gaps=gaps), **kw)
        want_wps = engine.wps(kind, a, b, CONTIG_LEN, W, 100, 200, 20)
        o = outs()
        w = torch.full((b - a,), -99, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()  # (torch fills the outputs on ITS stream, the engine launches on its own)
        engine.window_features_wps(kind, ws, we, w, a, b, CONTIG_LEN, coverage=o["coverage"], hist=o["hist"],
                                   hist_bins=(20, 640), overflow=o["overflow"], delfi_q=30, bl_start=bl_s, bl_end=bl_e,
                                   gaps=gaps, short=o["short"], long=o["long"], window_size=W, wps_min_length=100,
                                   wps_max_length=200, wps_quality=20, **kw)
        torch.cuda.synchronize()
        assert np.array_equal(w.cpu().numpy(), want_wps), kw
        for key in ("coverage", "overflow", "short", "long"):
            assert np.array_equal(o[key].cpu().numpy(), want[key]), (key, kw)
        assert np.array_equal(o["hist"].cpu().numpy().astype(np.uint32), want["hist"]), kw
    # one feature at a time, and a host WPS array (the merged launch, its scores copied back by the library)
    o = outs()
    w = torch.empty(CONTIG_LEN, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()  # (torch fills the outputs on ITS stream, the engine launches on its own)
    engine.window_features_wps(kind, ws, we, w, 0, CONTIG_LEN, CONTIG_LEN, coverage=o["coverage"])
    engine.sync()
    assert np.array_equal(o["coverage"].cpu().numpy(), engine.window_counts(kind, ws, we, 30))
    engine.window_features_wps(kind, ws, we, w, 0, CONTIG_LEN, CONTIG_LEN, delfi_q=30, bl_start=bl_s, bl_end=bl_e, gaps=gaps,
                               short=o["short"], long=o["long"])
    engine.sync()
    sh, lg, _ = engine.delfi_counts(kind, ws, we, 30, bl_s, bl_e, gaps)
    assert np.array_equal(o["short"].cpu().numpy(), sh) and np.array_equal(o["long"].cpu().numpy(), lg)
    assert np.array_equal(w.cpu().numpy(), engine.wps(kind, 0, CONTIG_LEN, CONTIG_LEN))
    host = np.zeros(CONTIG_LEN, np.int64)
    cov = np.zeros(n_win, np.int64)
    engine.window_features_wps(kind, ws, we, host, 0, CONTIG_LEN, CONTIG_LEN, coverage=cov)
    assert np.array_equal(host, engine.wps(kind, 0, CONTIG_LEN, CONTIG_LEN))
    assert np.array_equal(cov, engine.window_counts(kind, ws, we, 30))


def test_wps_host_results_cross_the_link_narrow_and_arrive_exact(engine):
    """Results of 4 M positions or more go to the host as int16 and are widened there (`copy_scores_narrow`,
    csrc/ftk_api.hip); a score beyond 16 bits - a pile of 40 000 identical fragments - must send the result the plain
    way.  Both against the same launch left on the device (which never takes that path) and against the closed form at
    the pile (reference frag/_wps.py:25-53: +1 per spanning fragment, -1 per fragment end inside the window)."""
    import torch
    size = 6_000_000
    rng = np.random.default_rng(5)
    s = np.sort(rng.integers(0, size - 400, 20_000)).astype(np.int32)
    e = (s + rng.integers(120, 181, len(s))).astype(np.int32)
    for name, pile in (("narrow_ok", 300), ("narrow_misfit", 40_000)):
        ps = np.full(pile, 3_000_000, np.int32)
        order = np.argsort(np.concatenate([s, ps]), kind="stable")
        S = np.concatenate([s, ps])[order]
        E = np.concatenate([e, ps + 150])[order]
        q = np.full(len(S), 60, np.uint8)
        engine.load_contig(name, S, E, q, np.ones(len(S), np.uint8))
        host = engine.wps(name, 0, size, size)
        dev = torch.empty(size, dtype=torch.int64, device="cuda:0")
        engine.wps(name, 0, size, size, out=dev)
        engine.sync()
        assert host.dtype == np.int64 and np.array_equal(host, dev.cpu().numpy()), name
        # bases 3 000 061 .. 3 000 090 are spanned by every fragment of the pile (and see none of its ends)
        assert int(host[3_000_061:3_000_091].min()) >= pile - 40 and int(host.max()) <= pile + 40, name
        assert (int(host.max()) > 32767) == (pile > 32767)
        # a short interval takes the plain copy: same numbers
        assert np.array_equal(engine.wps(name, 2_990_000, 3_010_000, size), host[2_990_000:3_010_000])
        # the merged launch (feature blocks + WPS tiles) with a HOST score array takes the same wire
        ws, we = synth.tiling_windows(size, 10_000)
        cov, merged_host = np.zeros(len(ws), np.int64), np.full(size, -1, np.int64)
        engine.window_features_wps(name, ws, we, merged_host, 0, size, size, coverage=cov)
        assert np.array_equal(merged_host, host) and np.array_equal(cov, engine.window_counts(name, ws, we, 30)), name
        # the host threads fill such a result, so it lives in ordinary memory (ftk_host_alloc_pageable: nothing to
        # page-lock in a process's first call) that the library recycles: the block of a dropped result serves the next
        where, keep = host.ctypes.data, host.copy()
        assert where % (2 << 20) == 0
        del host
        again = engine.wps(name, 0, size, size)
        assert again.ctypes.data == where and np.array_equal(again, keep), name
        # ... and into a page-locked array of the caller's (the round-4 default before) the numbers are the same
        pinned = engine.result_array(size, np.int64)
        assert pinned.ctypes.data != where and np.array_equal(engine.wps(name, 0, size, size, out=pinned), keep)
        del again, pinned
        engine.release(name)
