"""adjust_wps: oracle vs the reference's recorded outputs (CPU) and the HIP path vs both (GPU).

Tolerances: the median path without the Savitzky-Golay pass is exact (selection + one
average); the mean path and the Savitzky-Golay pass are float64 sums whose order differs
from numpy's pairwise / scipy's edge polyfit, so they are compared at rtol=1e-9, atol=1e-9
(the bigWig container stores float32, i.e. ~1e-7 relative)."""
import json
import os

import numpy as np
import pytest

from oracle import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
TOL = dict(rtol=1e-9, atol=1e-9)


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "adjust_wps.npz")), json.load(open(os.path.join(GOLD, "adjust_wps.json")))


def track_runs(A):
    runs = {}
    for k in A.files:
        if k.startswith("track_") and k.endswith("_start"):
            _, c, i, _ = k.split("_")
            runs.setdefault(c, []).append((int(A[k]), A[f"track_{c}_{i}_values"]))
    return runs


def raw_slice(runs, contig, start, n):
    for s0, v in runs[contig]:
        if s0 <= start and start + n <= s0 + len(v):
            # pyBigWig hands back float32-precision values
            return v[start - s0:start - s0 + n].astype(np.float32).astype(np.float64)
    raise AssertionError("run not inside the track")


def case_inputs(A, cs, i):
    """The raw scores behind output run ``i`` of a golden case (run start - W/2, length + W)."""
    W = cs.get("median_window_size", 1000)
    vals = A[f"{cs['key']}_{i}_values"]
    start = int(A[f"{cs['key']}_{i}_start"]) - W // 2
    return raw_slice(track_runs(A), cs["run_contigs"][i], start, len(vals) + W), vals, W


def run_kwargs(cs, W):
    return dict(window_size=W, use_mean=cs.get("mean", False),
                edge_size=cs.get("edge_size", 500) if cs.get("subtract_edges") else None,
                savgol_window=cs.get("savgol_window_size", 21), savgol_deg=cs.get("savgol_poly_deg", 2),
                savgol=cs.get("savgol", True))


def test_oracle_matches_reference_outputs(gold):
    A, cases = gold
    for cs in cases:
        assert cs["n_runs"] >= 3
        for i in (0, cs["n_runs"] - 1):
            raw, want, W = case_inputs(A, cs, i)
            got = O.py_adjust_run(raw, **run_kwargs(cs, W))
            if cs.get("mean"):
                np.testing.assert_allclose(got, want, **TOL)
            else:
                assert np.array_equal(got, want), cs["key"]
    x = A["float_input"]
    assert np.array_equal(O.py_adjust_run(x, 400, savgol=False), A["float_median400"])
    np.testing.assert_allclose(O.py_adjust_run(x, 400, use_mean=True, savgol=False), A["float_mean400"], **TOL)


def test_interval_merge_rules(tmp_path):
    from finaletoolkit_amd.frag._adjust_wps import _read_intervals
    bed = os.path.join(GOLD, "adjust_sites.bed")
    iv = _read_intervals(bed, 5000, 1000)
    # 42500 and 45050 +/- 2500 overlap after trimming 500 from each end -> merged
    assert ("chrA", 40000, 47550) in iv and ("chrA", 0, 3600) in iv
    iv3 = _read_intervals(bed, 3000, 1000)
    assert ("chrA", 41000, 44000) in iv3 and ("chrA", 43550, 46550) in iv3
    with pytest.raises(ValueError):
        _read_intervals(str(tmp_path / "x.txt"), 5000, 1000)


# ---------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_gpu_float_median_exact(engine, gold):
    A, _ = gold
    x = A["float_input"]
    offs = np.array([0, len(x)], np.int64)
    got = engine.wps_adjust(x, offs, 400, savgol=False)
    assert np.array_equal(got, A["float_median400"])
    got = engine.wps_adjust(x, offs, 400, mean=True, savgol=False)
    np.testing.assert_allclose(got, A["float_mean400"], **TOL)


@pytest.mark.gpu
def test_gpu_cases_match_reference(engine, gold):
    A, cases = gold
    for cs in cases:
        W = cs.get("median_window_size", 1000)
        raws, wants = [], []
        for i in range(cs["n_runs"]):
            raw, want, _ = case_inputs(A, cs, i)
            raws.append(raw)
            wants.append(want)
        offs = np.zeros(len(raws) + 1, np.int64)
        np.cumsum([len(r) for r in raws], out=offs[1:])
        sub = None
        if cs.get("subtract_edges"):
            e = cs.get("edge_size", 500)
            sub = [np.mean([np.mean(r[:e]), np.mean(r[-e:])]) for r in raws]
        got = engine.wps_adjust(np.concatenate(raws), offs, W, cs.get("mean", False), sub,
                                cs.get("savgol_window_size", 21), cs.get("savgol_poly_deg", 2),
                                cs.get("savgol", True))
        want = np.concatenate(wants)
        assert got.shape == want.shape
        if not cs.get("savgol", True) and not cs.get("mean"):
            assert np.array_equal(got, want), cs["key"]
        else:
            np.testing.assert_allclose(got, want, err_msg=cs["key"], **TOL)


@pytest.mark.gpu
@pytest.mark.parametrize("W", [2, 64, 1000, 2048])
def test_gpu_ragged_runs_vs_oracle(engine, W):
    rng = np.random.default_rng(W)
    lens = [W + 21, W + 22, 3 * W + 50, W + 4000, W + 1049, W + 1050]
    runs = [np.round(rng.normal(0, 30, n)) for n in lens]
    runs[2] = rng.normal(0, 1, lens[2])              # non-integer
    runs[3][:] = 7.0                                  # all ties
    offs = np.zeros(len(runs) + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    flat = np.concatenate(runs)
    got = engine.wps_adjust(flat, offs, W, savgol=False)
    want = np.concatenate([O.py_adjust_run(r, W, savgol=False) for r in runs])
    assert np.array_equal(got, want)
    got = engine.wps_adjust(flat, offs, W, savgol=True)
    want = np.concatenate([O.py_adjust_run(r, W) for r in runs])
    np.testing.assert_allclose(got, want, **TOL)
    got = engine.wps_adjust(flat, offs, W, mean=True, savgol_window_size=11, savgol_poly_deg=3)
    want = np.concatenate([O.py_adjust_run(r, W, use_mean=True, savgol_window=11, savgol_deg=3) for r in runs])
    np.testing.assert_allclose(got, want, **TOL)


@pytest.mark.gpu
@pytest.mark.parametrize("W", [2, 120, 1000, 2048])
def test_gpu_histogram_median_takes_what_it_can_and_leaves_the_rest(engine, W):
    """Integer scores in a narrow range go through the sliding-histogram median (`adjust_median_hist_kernel`), anything
    else - a range beyond 256, a non-integer, a -0.0 - is left to the sort kernel, interval by interval; both give the
    numpy median of the oracle bit for bit, with and without the edge term."""
    rng = np.random.default_rng(100 + W)
    lens = [W + 1, W + 63, W + 64, W + 4096, W + 4097, W + 9000, W + 5000, W + 3000, W + 700, W + 2000]
    runs = [rng.integers(-40, 25, n).astype(np.float64) for n in lens]
    runs[3] = np.round(rng.normal(0, 3, lens[3]))                      # many ties, few bins
    runs[5] = np.repeat(rng.integers(-100, 100, lens[5] // 50 + 1), 50)[:lens[5]].astype(np.float64)   # plateaus: pointers jump
    runs[6][2500] = 400.0                                              # range beyond 256 in ONE tile of the interval
    runs[7][10] = 0.5                                                  # a non-integer
    runs[8][5] = -0.0                                                  # negative zero
    runs[9] = rng.integers(-(1 << 20), 1 << 20, lens[9]).astype(np.float64)   # wide integers
    offs = np.zeros(len(runs) + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    flat = np.concatenate(runs)
    got = engine.wps_adjust(flat, offs, W, savgol=False)
    want = np.concatenate([O.py_adjust_run(r, W, savgol=False) for r in runs])
    assert np.array_equal(got, want)
    e = max(1, min(500, W // 2))
    sub = [np.mean([np.mean(r[:e]), np.mean(r[-e:])]) for r in runs]
    got = engine.wps_adjust(flat, offs, W, False, sub, savgol=False)
    want = np.concatenate([O.py_adjust_run(r, W, edge_size=e, savgol=False) for r in runs])
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_gpu_many_short_runs_build_their_tiles_on_the_device(engine):
    """40 000 runs (more than one trip of the tile-count kernel's 16 384) of W .. W + 150 scores - runs with NO output
    (length == W) among them, integer and non-integer runs interleaved so that every kernel of the chain has tiles:
    every 97th run against the oracle, the rest through the negation property."""
    rng = np.random.default_rng(4242)
    n_iv, W = 40_000, 64
    lens = W + rng.integers(0, 151, n_iv)
    lens[::11] = W                                     # nothing to write for these
    offs = np.zeros(n_iv + 1, np.int64)
    np.cumsum(lens, out=offs[1:])
    x = rng.integers(-30, 30, int(offs[-1])).astype(np.float64)
    for i in range(5, n_iv, 13):                       # non-integers: left to the sort kernel
        x[offs[i]] += 0.5
    for i in range(7, n_iv, 17):                       # a range beyond 128 but within 256: the second histogram pass
        if lens[i] > W + 2:
            x[offs[i] + 1] = 150.0
    got = engine.wps_adjust(x, offs, W, savgol=False)
    out_offs = offs - np.arange(n_iv + 1) * W
    assert len(got) == out_offs[-1]
    for i in range(0, n_iv, 97):
        want = O.py_adjust_run(x[offs[i]:offs[i + 1]], W, savgol=False) if lens[i] > W else np.zeros(0)
        assert np.array_equal(got[out_offs[i]:out_offs[i + 1]], want), i
    assert np.array_equal(engine.wps_adjust(-x, offs, W, savgol=False), -got)


@pytest.mark.gpu
def test_gpu_adjust_errors(engine):
    x = np.zeros(1500)
    offs = np.array([0, 1500], np.int64)
    with pytest.raises(ValueError):
        engine.wps_adjust(x, offs, 2000)                 # window longer than the run
    with pytest.raises(ValueError):
        engine.wps_adjust(x, offs, 1001)                 # odd window: numpy broadcast error in the reference
    with pytest.raises(ValueError):
        engine.wps_adjust(x, offs, 1490, savgol_window_size=21)   # 10 filtered values < 21
    assert len(engine.wps_adjust(x, offs, 1500, savgol=False)) == 0


@pytest.mark.gpu
def test_gpu_adjust_wps_file_to_file(engine, gold, tmp_path):
    from finaletoolkit_amd import frag
    from finaletoolkit_amd.bigwig import BigWigFile, write_fixed_step_bigwig
    A, cases = gold
    runs = track_runs(A)
    sizes = os.path.join(GOLD, "adjust.chrom.sizes")
    header = [(l.split()[0], int(l.split()[1])) for l in open(sizes)]
    raw_bw = str(tmp_path / "raw.bw")
    write_fixed_step_bigwig(raw_bw, header, [(c, s0, v) for c, _ in header for s0, v in sorted(runs[c])])
    for cs in cases:
        out_bw = str(tmp_path / f"{cs['key']}.bw")
        kw = {k: v for k, v in cs.items() if k not in ("key", "n_runs", "run_contigs")}
        bed = os.path.join(GOLD, "adjust_sites_5k.bed" if cs["key"] == "merge5k" else "adjust_sites.bed")
        frag.adjust_wps(raw_bw, bed, out_bw, sizes, **kw)
        bw = BigWigFile(out_bw)
        assert bw.chroms() == dict(header)
        for i in range(cs["n_runs"]):
            want = A[f"{cs['key']}_{i}_values"].astype(np.float32)
            s = int(A[f"{cs['key']}_{i}_start"])
            st, en, v = bw.intervals(cs["run_contigs"][i], s, s + len(want))
            assert st[0] == s and len(st) == len(want) and np.all(np.diff(st) == 1)
            np.testing.assert_allclose(v.astype(np.float32), want, rtol=1e-6, atol=1e-6, err_msg=cs["key"])


@pytest.mark.gpu
def test_gpu_adjust_full_size_properties(engine):
    """2 000 intervals x 5 kb (10 M scores): a sample of intervals against the oracle, and two
    size-independent properties of the median path on integer-valued data -- adding a constant to the
    input leaves the output unchanged, negating the input negates it (the window median of an even
    window is the mean of the two middle values, so both are exact)."""
    rng = np.random.default_rng(123)
    n_iv, ilen, W = 2000, 5000, 1000
    x = np.round(rng.normal(0, 40, n_iv * ilen))
    offs = np.arange(n_iv + 1, dtype=np.int64) * ilen
    got = engine.wps_adjust(x, offs, W, savgol=False)
    m = ilen - W
    for i in (0, 1, 777, n_iv - 1):
        assert np.array_equal(got[i * m:(i + 1) * m], O.py_adjust_run(x[i * ilen:(i + 1) * ilen], W, savgol=False)), i
    assert np.array_equal(engine.wps_adjust(x + 1234.0, offs, W, savgol=False), got)
    assert np.array_equal(engine.wps_adjust(-x, offs, W, savgol=False), -got)
    sg = engine.wps_adjust(x, offs, W)
    i = 1500
    np.testing.assert_allclose(sg[i * m:(i + 1) * m], O.py_adjust_run(x[i * ilen:(i + 1) * ilen], W), **TOL)
