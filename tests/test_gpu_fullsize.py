"""
GPU, BASELINE.json full sizes (configs 2 and 3: synthetic chr22 at 30x = 5 130 457
fragments): bit-exact against the C oracle where it finishes in seconds, and
size-independent properties for the rest.
"""
import numpy as np
import pytest

from finaletoolkit_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu

CHR22 = synth.B37_SIZES["22"]


@pytest.fixture(scope="module")
def chr22(engine):
    s, e, q, st = synth.synth_contig(CHR22, depth=30.0, seed=synth.SEED_BASE + 21)
    assert len(s) == 5_130_457
    engine.load_contig("chr22_30x", s, e, q, st)
    yield dict(s=s, e=e, q=q, st=st, fr=O.Frags(s, e, q, st))
    engine.release("chr22_30x")


def test_config2_coverage_and_histogram_bit_exact(engine, chr22):
    ws, we = synth.tiling_windows(CHR22, 100_000)
    assert len(ws) == 514
    cov = engine.window_counts("chr22_30x", ws, we, quality_threshold=30)
    assert np.array_equal(cov, O.c_window_counts(chr22["fr"], ws, we, mapq_min=30))
    hist, over = engine.fraglen_hist("chr22_30x", ws, we, 0, 1001, quality_threshold=30)
    h2, o2 = O.c_fraglen_hist(chr22["fr"], ws, we, 0, 1001, mapq_min=30)
    assert np.array_equal(hist, h2) and np.array_equal(over, o2)
    # properties: every mapq>=30 fragment lands in exactly one tiling window / one histogram bin
    n_pass = int((chr22["q"] >= 30).sum())
    assert cov.sum() == n_pass and int(hist.sum()) + int(over.sum()) == n_pass
    assert np.array_equal(hist.sum(axis=1) + over, cov)
    lens = (chr22["e"] - chr22["s"])[chr22["q"] >= 30]
    assert np.array_equal(hist.sum(axis=0), np.bincount(lens, minlength=1001)[:1001])
    # 'any' policy counts a fragment once per window it touches: total = passes + boundary crossings
    cov_any = engine.window_counts("chr22_30x", ws, we, quality_threshold=30, intersect_policy="any")
    s, e = chr22["s"][chr22["q"] >= 30].astype(np.int64), chr22["e"][chr22["q"] >= 30].astype(np.int64)
    crossings = int(((e - 1) // 100_000 - s // 100_000).sum())
    assert cov_any.sum() == n_pass + crossings
    # 5 Mb bins (config 4 geometry) = sums of fifty 100 kb bins
    ws5, we5 = synth.tiling_windows(CHR22, 5_000_000)
    cov5 = engine.window_counts("chr22_30x", ws5, we5, quality_threshold=30)
    assert np.array_equal(cov5, np.add.reduceat(cov, np.arange(0, len(cov), 50)))


def test_config2_delfi_full(engine, chr22):
    ws, we = synth.tiling_windows(CHR22, 100_000)
    rng = np.random.default_rng(9)
    bl_s = np.sort(rng.integers(0, CHR22 - 6000, 300)).astype(np.int32)
    bl_e = (bl_s + rng.integers(200, 5000, 300)).astype(np.int32)
    o = np.lexsort((bl_e, bl_s))
    gaps = (13_000_000, 16_000_000, [(0, 10_000), (CHR22 - 10_000, CHR22)])
    got = engine.delfi_counts("chr22_30x", ws, we, 30, bl_s[o], bl_e[o], gaps)
    want = O.c_delfi_counts(chr22["fr"], ws, we, 30, bl_s[o], bl_e[o], gaps)
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    assert np.array_equal(got[0] + got[1], got[2])


def test_config3_wps_whole_contig(engine, chr22):
    whole = engine.wps("chr22_30x", 0, CHR22, CHR22, 120, 120, 180, 30)
    assert whole.dtype == np.int64 and len(whole) == CHR22
    # bit-exact against the oracle on sampled 5 kb tiles (the reference's multi_wps geometry)
    rng = np.random.default_rng(3)
    for a in [0, CHR22 - 5000] + [int(x) for x in rng.integers(0, CHR22 - 5000, 10)]:
        assert np.array_equal(whole[a:a + 5000], O.c_wps(chr22["fr"], a, a + 5000, CHR22, 120, 120, 180, 30)), a
    # one call == the 10 261 x 5 kb tiling (W <= max_len: the per-tile fetch cut drops nothing)
    starts = np.arange(0, CHR22, 5000)
    tiled, offs = engine.wps_intervals("chr22_30x", starts, np.minimum(starts + 5000, CHR22), CHR22, 120, 120, 180, 30)
    assert len(starts) == 10_261 and np.array_equal(tiled, whole)
    # closed form of the sum: a passing fragment adds (len - W) spanning bases and 2W end bases (len >= W),
    # clipped to the contig -> compare with a numpy difference-array evaluation of the same closed form
    keep = (chr22["q"] >= 30) & (chr22["e"] - chr22["s"] >= 120) & (chr22["e"] - chr22["s"] <= 180)
    fs, fe = chr22["s"][keep].astype(np.int64), chr22["e"][keep].astype(np.int64)
    d = np.zeros(CHR22 + 400, np.int64)
    off = 100
    for pos, val in ((fs - 59, -1), (fs + 61, 2), (fe - 59, -2), (fe + 61, 1)):
        np.add.at(d, pos + off, val)
    ref = np.cumsum(d)[off:off + CHR22]
    assert np.array_equal(whole, ref)


def test_next_rows_at_full_size(engine, chr22, tmp_path):
    """chr22 at 30x through the next-row kernels: cleavage against oracle tiles, and the end-motif pass
    against size-independent identities (no oracle run over 5 M fragments)."""
    s, e, q = chr22["s"].astype(np.int64), chr22["e"].astype(np.int64), chr22["q"]
    fr, size, name = chr22["fr"], CHR22, "chr22_30x"
    prop = engine.cleavage(name, 0, size, None, None, 20)
    for a in (0, 17_000_123, 33_333_333, size - 4_000):
        assert np.array_equal(prop[a:a + 4_000], O.c_cleavage(fr, a, a + 4_000, None, None, 20)[2]), a
    # motifs on a random genome with one N stripe
    rng = np.random.default_rng(77)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size)].copy()
    n_lo, n_hi = 20_000_000, 20_000_500
    seq[n_lo:n_hi] = ord("N")
    from tests import helpers as H
    H.write_2bit(tmp_path / "chr22.2bit", {"chr22": seq.tobytes().decode()})
    from finaletoolkit_amd.reference import ReferenceGenome
    ws = np.arange(0, size, 1_000_000, dtype=np.int64)
    we = np.minimum(ws + 1_000_000, size)
    k = 4
    with ReferenceGenome(str(tmp_path / "chr22.2bit")) as ref:
        rid = ref.device_image(engine, "chr22")
        both, nf, err = engine.motif_counts(name, rid, ws, we, k, 0, -k, True, False, 0, False, 30)
        rev, _, _ = engine.motif_counts(name, rid, ws, we, k, 0, -k, False, True, 0, False, 30)
    assert err.sum() == 0
    ok = q >= 30
    # a fragment is fetched once per window it overlaps: twice when it crosses a 1 Mb boundary
    edges = ws[1:]
    idx = np.searchsorted(edges, s, side="right")                 # first boundary > start
    crosses = (idx < len(edges)) & (edges[np.minimum(idx, len(edges) - 1)] < e)
    w = 1 + crosses.astype(np.int64)
    assert nf.sum() == int((w * ok).sum())
    fwd_clean = ~((s + k > n_lo) & (s < n_hi))                    # 5' k-mer free of N
    rev_clean = ~((e > n_lo) & (e - k < n_hi))                    # 3' k-mer free of N
    assert int(rev.sum()) == int((w * (ok & rev_clean)).sum())
    assert int(both.sum()) == int((w * (ok & fwd_clean)).sum()) + int((w * (ok & rev_clean)).sum())
    # every 4-mer of a uniform random genome shows up about equally often
    tot = both.sum(axis=0)
    assert tot.min() > 0.9 * tot.mean() and tot.max() < 1.1 * tot.mean()
