"""
Pins the oracle (oracle/ftk_oracle.c and the pure-Python restatement) to the
reference: every vector in tests/golden/ was produced by importing the
reference itself (oracle/gen_golden.py, build container only).
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import DATA, GOLDEN, golden_json, golden_npz, read_bed, read_frag_gz

FIX = os.path.join(DATA, "12.3444.b37.frag.gz")
SYN = os.path.join(GOLDEN, "synth.frag.gz")


@pytest.fixture(scope="module")
def G():
    return golden_json()


@pytest.fixture(scope="module")
def A():
    return golden_npz()


@pytest.fixture(scope="module")
def fix():
    s, e, q, st = read_frag_gz(FIX)["12"]
    return dict(fr=O.Frags(s, e, q, st), rows=list(zip(s.tolist(), e.tolist(), q.tolist(), st.tolist())))


@pytest.fixture(scope="module")
def syn():
    d = read_frag_gz(SYN)
    return {c: dict(fr=O.Frags(*v), rows=list(zip(*[x.tolist() for x in v]))) for c, v in d.items()}


def test_fixture_rows_match_reference_stream(G, fix):
    want = [(r[1], r[2], r[3], int(r[4])) for r in G["fixture"]["frag_generator_all"]]
    assert fix["rows"] == want
    assert [(r[1], r[2], r[3], int(r[4])) for r in G["fixture"]["frag_generator_bed6"]] == want
    bed6 = read_frag_gz(os.path.join(DATA, "12.3444.b37.frag.bed.gz"))["12"]
    assert list(zip(*[x.tolist() for x in bed6])) == want


def test_fixture_coverage(G, fix):
    for c in G["fixture"]["single_coverage"]:
        kw = dict(mapq_min=c["q"], min_len=c["min_length"], max_len=c["max_length"], policy=c["policy"])
        got = O.c_window_counts(fix["fr"], [c["start"]], [c["stop"]], **kw)[0]
        assert got == c["coverage"], c
        gotp = O.py_single_coverage(fix["rows"], c["start"], c["stop"], c["min_length"], c["max_length"],
                                    c["policy"], c["q"])
        assert gotp == c["coverage"], c
    iv = read_bed(os.path.join(DATA, "intervals.bed"))
    got = O.c_window_counts(fix["fr"], [i[1] for i in iv], [i[2] for i in iv], mapq_min=30)
    assert got.tolist() == [r[4] for r in G["fixture"]["coverage_raw"]] == [4, 7]
    total = O.c_window_counts(fix["fr"], [0], [None], mapq_min=30)[0]
    assert total == 16
    for g, w in zip(got, G["fixture"]["coverage_norm"]):
        assert g * (1.0 / total) == w[4]


def test_fixture_select_and_lengths(G, fix):
    s, e, q, st = O.c_frag_select(fix["fr"], 34443119, 34443538, mapq_min=30)
    want = G["fixture"]["frag_generator_detail"]
    assert list(zip(s.tolist(), e.tolist(), q.tolist(), st.tolist())) == [(r[1], r[2], r[3], int(r[4])) for r in want]
    s, e, q, st = O.c_frag_select(fix["fr"], 34443119, 34443538, mapq_min=30, min_len=0, max_len=1000000000)
    assert (e - s).tolist() == G["fixture"]["frag_length"] == [166, 161, 197, 180]
    s, e, q, st = O.c_frag_select(fix["fr"], None, None, mapq_min=30, min_len=120, max_len=180)
    assert [[a, b, bool(c)] for a, b, c in zip(s.tolist(), e.tolist(), st.tolist())] == G["fixture"]["frag_array_120_180"]


def test_fixture_wps(G, A, fix):
    assert G["fixture"]["wps_145_155"] == [-1, -1, -1, -1, -1, 1, 1, 1, 1, 1]
    got = O.c_wps(fix["fr"], 34444145, 34444155, 133851895, 120, 120, 180, 0)
    assert got.tolist() == G["fixture"]["wps_145_155"]
    assert O.py_wps(fix["rows"], 34444145, 34444155, 133851895, quality_threshold=0).tolist() == G["fixture"]["wps_145_155"]
    got = O.c_wps(fix["fr"], 34443000, 34447000, 133851895, 120, 120, 180, 0)
    assert np.array_equal(got, A["fixture_wps_34443000_34447000"])


def _stats_from_hist(hist_row, len_lo, short_reads):
    nz = np.nonzero(hist_row)[0]
    dist = {int(b + len_lo): int(hist_row[b]) for b in nz}
    return O.py_frag_length_stats(dist, short_reads)


def test_fixture_frag_length_stats(G, fix):
    assert G["fixture"]["median_quirk"] == 147.0
    assert O.py_find_median({121: 1, 137: 1, 147: 1, 152: 1, 161: 1, 170: 1, 205: 1}) == 147.0
    iv = read_bed(os.path.join(DATA, "intervals.bed"))
    hist, over = O.c_fraglen_hist(fix["fr"], [i[1] for i in iv], [i[2] for i in iv], 0, 1001, mapq_min=30, min_len=0)
    assert over.sum() == 0
    for i, want in enumerate(G["fixture"]["frag_length_intervals"]):
        got = _stats_from_hist(hist[i], 0, 150)
        assert got[1] == want[5] and got[3:6] == tuple(want[7:10])
        assert got[0] == pytest.approx(want[4], rel=1e-12) and got[2] == pytest.approx(want[6], rel=1e-9)
        assert got[6] == pytest.approx(want[10], rel=1e-12)
        d = O.py_distribution(fix["rows"], iv[i][1], iv[i][2], 0, None, "midpoint", 30)
        assert O.py_frag_length_stats(d, 150) == tuple(want[4:])


def test_synth_coverage_all_variants(G, syn):
    wins = read_bed(os.path.join(GOLDEN, "synth_windows.bed"))
    variants = {
        "default": dict(mapq_min=30),
        "any_q0": dict(mapq_min=0, policy="any"),
        "len_120_180": dict(mapq_min=30, min_len=120, max_len=180),
        "q60_max150_any": dict(mapq_min=60, max_len=150, policy="any"),
        "min300": dict(mapq_min=10, min_len=300),
    }
    for key, kw in variants.items():
        want = G["synth"]["coverage"][key]
        assert [tuple(w[:4]) for w in want] == wins
        for contig in syn:
            idx = [i for i, w in enumerate(wins) if w[0] == contig]
            got = O.c_window_counts(syn[contig]["fr"], [wins[i][1] for i in idx], [wins[i][2] for i in idx], **kw)
            assert got.tolist() == [want[i][4] for i in idx], key
    # pure-Python restatement on a subset (slow loops)
    want = G["synth"]["coverage"]["q60_max150_any"]
    for i in list(range(0, len(wins), 7)):
        c, a, b, _ = wins[i]
        assert O.py_single_coverage(syn[c]["rows"], a, b, None, 150, "any", 60) == want[i][4]
    whole = sum(O.c_window_counts(syn[c]["fr"], [0], [None], mapq_min=30)[0] for c in syn)
    assert whole == G["synth"]["single_coverage_whole_file"]
    assert O.c_window_counts(syn["chrA"]["fr"], [0], [None], mapq_min=30)[0] == G["synth"]["single_coverage_whole_chrA"]
    for g, w in zip(G["synth"]["coverage"]["default"], G["synth"]["coverage"]["normalized"]):
        assert g[4] * (1e6 / whole) == w[4]


def test_synth_frag_length(G, A, syn):
    wins = read_bed(os.path.join(GOLDEN, "synth_windows.bed"))
    for key, kw, short in [("frag_length_intervals", dict(mapq_min=30, min_len=0), 150),
                           ("frag_length_intervals_120_400_any", dict(mapq_min=0, min_len=120, max_len=400, policy="any"), 167)]:
        want = G["synth"][key]
        for contig in syn:
            idx = [i for i, w in enumerate(wins) if w[0] == contig]
            hist, over = O.c_fraglen_hist(syn[contig]["fr"], [wins[i][1] for i in idx], [wins[i][2] for i in idx],
                                          0, 1001, **kw)
            assert over.sum() == 0
            for j, i in enumerate(idx):
                got = _stats_from_hist(hist[j], 0, short)
                w = want[i]
                assert got[1] == w[5] and list(got[3:6]) == w[7:10], (key, i)
                assert got[0] == pytest.approx(w[4], rel=1e-12) and got[2] == pytest.approx(w[6], rel=1e-9)
                assert got[6] == pytest.approx(w[10], rel=1e-12)
    s, e, _, _ = O.c_frag_select(syn["chrB"]["fr"], 10_000, 30_000, mapq_min=0, min_len=0, max_len=1000000000, policy="any")
    assert np.array_equal(e - s, A["synth_frag_length_chrB_any"])
    s, e, _, _ = O.c_frag_select(syn["chrA"]["fr"], None, None, mapq_min=30, min_len=0, max_len=1000000000)
    assert np.array_equal(e - s, A["synth_frag_length_chrA_all"])
    s, e, q, st = O.c_frag_select(syn["chrA"]["fr"], 100_000, 130_000, mapq_min=20, min_len=100, max_len=400, policy="any")
    want = G["synth"]["frag_generator_chrA_any"]
    assert list(zip(s.tolist(), e.tolist(), q.tolist(), st.tolist())) == [(r[1], r[2], r[3], int(r[4])) for r in want]
    # binned distribution of a whole contig (bins/counts as frag_length_bins returns them)
    hist, _ = O.c_fraglen_hist(syn["chrA"]["fr"], [None], [None], 0, 1001, mapq_min=30, min_len=0)
    want = G["synth"]["frag_length_bins_chrA"]
    lo, hi = want["bins"][0], want["bins"][-1]
    assert hist[0][lo:hi + 1].tolist() == want["counts"] and hist[0].sum() == sum(want["counts"])


def test_synth_wps(G, A, syn):
    for c in G["synth"]["wps_cases"]:
        want = A[c["key"]]
        got = O.c_wps(syn[c["contig"]]["fr"], c["start"], c["stop"], G["synth"]["contigs"][c["contig"]],
                      c["window_size"], c["min_length"], c["max_length"], c["quality_threshold"])
        assert np.array_equal(got, want), c
    c = G["synth"]["wps_cases"][3]
    got = O.py_wps(syn[c["contig"]]["rows"], c["start"], c["stop"], G["synth"]["contigs"][c["contig"]],
                   c["window_size"], c["min_length"], c["max_length"], c["quality_threshold"])
    assert np.array_equal(got, A[c["key"]])


def test_synth_delfi_windows(G, syn):
    bl = {}
    for c, a, b, _ in read_bed(os.path.join(GOLDEN, "synth_blacklist.bed")):
        bl.setdefault(c, []).append((a, b))
    gaps = {k: (v["centromere"][0], v["centromere"][1], [tuple(t) for t in v["telomeres"]])
            for k, v in G["synth"]["gaps"].items()}
    rows = [r for r in G["synth"]["delfi_windows"] if r["arm"] != "NOARM"]
    assert rows
    for use_gaps in (True, False):
        for use_bl in (True, False):
            for contig in syn:
                sel = [r for r in rows if r["gaps"] == use_gaps and r["blacklist"] == use_bl and r["contig"] == contig]
                if not sel:
                    continue
                b = sorted(bl[contig]) if use_bl else []
                sh, lg, nf = O.c_delfi_counts(syn[contig]["fr"], [r["start"] for r in sel], [r["stop"] for r in sel], 30,
                                              [x[0] for x in b] or None, [x[1] for x in b] or None,
                                              gaps[contig] if use_gaps else None)
                assert sh.tolist() == [r["short"] for r in sel]
                assert lg.tolist() == [r["long"] for r in sel]
                assert nf.tolist() == [r["num_frags"] for r in sel]
                for r in sel[::5]:
                    got = O.py_delfi_single_window(syn[contig]["rows"], r["start"], r["stop"], 30, b,
                                                   gaps[contig] if use_gaps else None)
                    assert got == (r["short"], r["long"], r["num_frags"])


def test_synth_cleavage(G, A, syn):
    for c in G["synth"]["cleavage_cases"]:
        size = G["synth"]["contigs"][c["contig"]]
        a, b = max(c["start"] - c["left"], 0), min(c["stop"] + c["right"], size)
        depth, ends, prop = O.c_cleavage(syn[c["contig"]]["fr"], a, b, c["min_length"], c["max_length"],
                                         c["quality_threshold"])
        assert np.array_equal(A["cleavage_pos_" + c["key"].split("_")[1]], np.arange(a, b))
        assert np.array_equal(prop, A[c["key"]]), c  # float64, bit for bit
