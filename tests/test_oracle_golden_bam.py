"""
Pins the oracle's BAM (read1-fetch) mode to the reference's OWN code: every vector in tests/golden/bam.json.gz /
bam.npz was produced by the imported reference (io/alignment.py `_fetch_sam`, utils/_frag_generator.py,
frag/_coverage.py, _wps.py, _frag_length.py, _delfi.py, _cleavage_profile.py) running over a BAM stand-in for pysam
(oracle/gen_golden_bam.py, oracle/bamstub.py; build container only) on the reference's 48-record fixture and on
tests/golden/edge.bam - multi-op CIGARs, CIGAR-less records, alignments that consume no reference, every rejected flag,
read1 outside the window that holds the midpoint, negative fragment starts.

Held to them here: `oracle.bam_rows` (the record rule), the C oracle (`orc_*` with read1 columns) and the pure-Python
restatement (`py_*`).
"""
import gzip
import json
import os

import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import DATA, GOLDEN

FIX = os.path.join(DATA, "12.3444.b37.bam")
EDGE = os.path.join(GOLDEN, "edge.bam")
NOCIGAR = os.path.join(GOLDEN, "edge_nocigar.bam")


def bam_golden():
    with gzip.open(os.path.join(GOLDEN, "bam.json.gz"), "rt") as fh:
        return json.load(fh)


@pytest.fixture(scope="module")
def G():
    return bam_golden()


@pytest.fixture(scope="module")
def A():
    return np.load(os.path.join(GOLDEN, "bam.npz"))


class Bam:
    def __init__(self, path):
        self.names, self.lengths, self.rows = O.bam_rows(path)
        self.fr, self.rank = {}, {}
        for c, r in self.rows.items():
            self.fr[c], self.rank[c] = O.frags_from_bam_rows(r) if r else (O.Frags([], [], [], [], [], []), np.zeros(0, np.int64))


@pytest.fixture(scope="module")
def bams():
    return {"fixture": Bam(FIX), "edge": Bam(EDGE), "negative_start": Bam(EDGE)}


SECTIONS = ["fixture", "edge", "negative_start"]


def _kw(kw):
    return dict(mapq_min=kw.get("quality_threshold", 30), min_len=kw.get("min_length"), max_len=kw.get("max_length"),
                policy=kw.get("intersect_policy", "midpoint"))


def _py(kw):
    return (kw.get("min_length"), kw.get("max_length"), kw.get("intersect_policy", "midpoint"), kw.get("quality_threshold", 30))


def test_header_and_record_rule(G, bams):
    """`bam_rows` = what the reference's AlignmentWrapper yields for whole contigs and for the whole file, in order."""
    for sec in SECTIONS:
        b = bams[sec]
        if sec != "negative_start":
            assert G[sec]["chroms"] == dict(zip(b.names, b.lengths)) and G[sec]["is_sam"] is True
        for case in G[sec]["fetch"]:
            if case["start"] is not None or case["stop"] is not None:
                continue
            q = case["quality_threshold"]
            contigs = b.names if case["contig"] is None else [case["contig"]]
            got = [[c, r[0], r[1], r[2], bool(r[3])] for c in contigs for r in b.rows[c] if r[2] >= q]
            assert got == case["fragments"], (sec, case["contig"], q)
    assert len(bams["fixture"].rows["12"]) == 17
    # the shapes the edge file exists for are really in it
    rows = {c: set(r[:2] for r in v) for c, v in bams["edge"].rows.items()}
    assert (120_000, 120_170) in rows["chrA"] and (9_990, 10_150) in rows["chrB"]          # CIGAR-less, TLEN > 0
    assert (129_851, 130_001) in rows["chrA"] and (131_000, 131_140) in rows["chrA"]       # no reference consumed
    assert {(-70, 50), (-1, 30), (-240, 160), (0, 25), (-149_750, 250)} <= rows["chrN"]    # negative starts
    assert any(r[4] < r[0] or r[5] > r[1] for r in bams["edge"].rows["chrA"])              # read1 poking out of its fragment


def test_region_fetch(G, bams):
    """AlignmentWrapper.fetch(contig, start, stop): read1 alignments overlapping the region, file order."""
    for sec in SECTIONS:
        b = bams[sec]
        for case in G[sec]["fetch"]:
            if case["start"] is None and case["stop"] is None:
                continue
            c, q = case["contig"], case["quality_threshold"]
            got = [[c, *r[:3], bool(r[3])] for r in O.py_fetch(b.rows[c], case["start"], case["stop"], q)]
            assert got == case["fragments"], (sec, c, case["start"], case["stop"], q)
            s, e, mq, st = O.c_frag_select(b.fr[c], case["start"], case["stop"], mapq_min=q, policy="fetch")
            assert sorted(zip(s.tolist(), e.tolist(), mq.tolist(), st.tolist())) == \
                sorted((r[1], r[2], r[3], int(r[4])) for r in case["fragments"])


def test_frag_generator_and_frag_array(G, bams):
    for sec in SECTIONS:
        b = bams[sec]
        for case in G[sec]["frag_generator"]:
            c = case["contig"]
            got = [[c, *r[:3], bool(r[3])] for r in O.py_frag_generator(b.rows[c], case["start"], case["stop"], *_py(case["kw"]))]
            assert got == case["fragments"], (sec, c, case["start"], case["stop"], case["kw"])
            s, e, mq, st = O.c_frag_select(b.fr[c], case["start"], case["stop"], **_kw(case["kw"]))
            assert sorted(zip(s.tolist(), e.tolist(), mq.tolist(), st.tolist())) == \
                sorted((r[1], r[2], r[3], int(r[4])) for r in case["fragments"])
        for case in G[sec]["frag_array"]:
            c = case["contig"]
            got = [[r[0], r[1], bool(r[3])] for r in O.py_frag_generator(b.rows[c], case["start"], case["stop"], *_py(case["kw"]))]
            assert got == case["rows"]


def test_single_coverage(G, bams):
    n = 0
    for sec in SECTIONS:
        b = bams[sec]
        by = {}
        for case in G[sec]["single_coverage"]:
            if case["contig"] is None:  # the whole file: the sum over its contigs
                tot = sum(O.py_single_coverage(b.rows[c], 0, None, *_py(case["kw"])) for c in b.names)
                assert tot == case["coverage"]
                continue
            by.setdefault((case["contig"], json.dumps(case["kw"], sort_keys=True)), []).append(case)
        for (c, _), cases in by.items():
            kw = cases[0]["kw"]
            got = O.c_window_counts(b.fr[c], [x["start"] for x in cases], [x["stop"] for x in cases], **_kw(kw))
            assert got.tolist() == [x["coverage"] for x in cases], (sec, c, kw)
            for x in cases[::7]:
                assert O.py_single_coverage(b.rows[c], x["start"], x["stop"], *_py(kw)) == x["coverage"]
            n += len(cases)
    assert n > 500
    fx = {(c["start"], c["stop"], json.dumps(c["kw"], sort_keys=True)): c["coverage"] for c in G["fixture"]["single_coverage"]}
    assert fx[(0, None, json.dumps({"quality_threshold": 0}))] == 17                   # reference tests/test_coverage.py
    assert fx[(34_443_400, 34_443_600, json.dumps({"quality_threshold": 0}))] == 2


def test_wps(G, A, bams):
    for sec in SECTIONS:
        b = bams[sec]
        sizes = dict(zip(b.names, b.lengths))
        for k, case in enumerate(G[sec]["wps"]):
            c = case["contig"]
            want = A[case["key"]]
            got = O.c_wps(b.fr[c], case["start"], case["stop"], sizes[c], case["window_size"], case["min_length"],
                          case["max_length"], case["quality_threshold"])
            assert np.array_equal(got, want), (sec, case)
            if case["stop"] - case["start"] <= 1500 and k % 3 == 0:
                gp = O.py_wps(b.rows[c], case["start"], case["stop"], sizes[c], case["window_size"], case["min_length"],
                              case["max_length"], case["quality_threshold"])
                assert np.array_equal(gp, want), (sec, case)
    assert A["fixture_wps_0"].tolist() == [-1, -1, -1, -1, -1, 1, 1, 1, 1, 1]          # reference tests/test_wps.py:18-26


def test_lengths_bins_and_intervals(G, A, bams):
    for sec in SECTIONS:
        b = bams[sec]
        for case in G[sec]["frag_length"]:
            c, kw = case["contig"], case["kw"]
            got = [r[1] - r[0] for r in O.py_frag_generator(b.rows[c], case["start"], case["stop"], 0, 1_000_000_000,
                                                            kw.get("intersect_policy", "midpoint"), kw.get("quality_threshold", 30))]
            assert got == A[case["key"]].tolist(), (sec, case)
        for case in G[sec]["frag_length_bins"]:
            kw = case["kw"]
            contigs = [kw["contig"]] if "contig" in kw else b.names
            mn, mx, bs = kw.get("min_length", 0), kw.get("max_length"), kw.get("bin_size", 1)
            q = kw.get("quality_threshold", 30)
            dist = {}
            for c in contigs:
                for ln, k in O.py_distribution(b.rows[c], None, None, mn, mx, "midpoint", q).items():
                    dist[ln] = dist.get(ln, 0) + k
            if not dist:  # frag/_frag_length.py: "No fragments found": empty arrays
                assert case["bins"] == [] and case["counts"] == []
                continue
            lo, hi = min(dist), max(dist)                                   # frag/_frag_length.py: bins span the observed lengths
            bins = np.arange(lo, hi + bs, bs)
            counts = np.zeros((hi - lo) // bs + 1, np.int64)
            for ln, k in dist.items():
                counts[(ln - lo) // bs] += k
            assert bins.tolist() == case["bins"] and counts.tolist() == case["counts"], (sec, kw)
    for sec, key, kw in (("fixture", "frag_length_intervals", {}), ("edge", "frag_length_intervals", {}),
                         ("edge", "frag_length_intervals_any_q0", dict(min_length=50, max_length=600, intersect_policy="any",
                                                                       quality_threshold=0, short_reads=167))):
        b = bams[sec]
        for row in G[sec][key]:
            c, a, z = row[0], row[1], row[2]
            dist = O.py_distribution(b.rows[c], a, z, kw.get("min_length"), kw.get("max_length"),
                                     kw.get("intersect_policy", "midpoint"), kw.get("quality_threshold", 30))
            got = O.py_frag_length_stats(dist, kw.get("short_reads", 150))
            assert got[5] == row[9] and got[3:5] == tuple(row[7:9]), (sec, row)
            for g, w in zip(got, row[4:]):
                assert g == pytest.approx(w, rel=1e-9), (sec, row)


def test_coverage_driver_rows(G, bams):
    """frag.coverage over the interval file, raw and normalised by the whole file's total (chrN's negative starts
    are part of that total)."""
    b = bams["edge"]
    for key, kw in (("coverage_default", {}), ("coverage_any_q0", dict(intersect_policy="any", quality_threshold=0)),
                    ("coverage_len_120_180", dict(min_length=120, max_length=180))):
        for row in G["edge"][key]:
            assert O.py_single_coverage(b.rows[row[0]], row[1], row[2], *_py(kw)) == row[4], (key, row)
    total = sum(O.py_single_coverage(b.rows[c], 0, None, None, None, "midpoint", 30) for c in b.names)
    for raw, norm in zip(G["edge"]["coverage_default"], G["edge"]["coverage_normalized"]):
        assert raw[4] * (1e6 / total) == norm[4]
    f = bams["fixture"]
    # (the frag.gz twin of the fixture gives 4, 7 of 16 - reference tests/test_coverage.py:45-89; its rows carry the
    #  pair's lower mapq, the BAM path sees read1's own: one more fragment passes the default cut of 30)
    iv = [(r[0], r[1], r[2]) for r in G["fixture"]["coverage_raw"]]
    want = [O.py_single_coverage(f.rows[c], a, z, None, None, "midpoint", 30) for c, a, z in iv]
    assert [r[4] for r in G["fixture"]["coverage_raw"]] == want == [4, 8]
    tot = O.py_single_coverage(f.rows["12"], 0, None, None, None, "midpoint", 30)
    assert tot == 17 and [r[4] for r in G["fixture"]["coverage_norm"]] == [4 * (1.0 / tot), 8 * (1.0 / tot)]


def test_delfi_windows(G, bams):
    b = bams["edge"]
    bl = {c: sorted(zip(*v)) for c, v in G["edge"]["blacklist"].items()}
    gaps = {c: (g["centromere"][0], g["centromere"][1], [tuple(t) for t in g["telomeres"]]) for c, g in G["edge"]["gaps"].items()}
    n = 0
    for use_gaps in (True, False):
        for use_bl in (True, False):
            for c in ("chrA", "chrB"):
                rows = [r for r in G["edge"]["delfi_windows"] if r["gaps"] == use_gaps and r["blacklist"] == use_bl and r["contig"] == c]
                keep = [r for r in rows if r["arm"] != "NOARM"]
                if not keep:
                    continue
                ws, we = [r["start"] for r in keep], [r["stop"] for r in keep]
                bs, be = ([x[0] for x in bl[c]], [x[1] for x in bl[c]]) if use_bl else (None, None)
                sh, lg, nf = O.c_delfi_counts(b.fr[c], ws, we, 30, bs, be, gaps[c] if use_gaps else None)
                assert sh.tolist() == [r["short"] for r in keep] and lg.tolist() == [r["long"] for r in keep]
                assert nf.tolist() == [r["num_frags"] for r in keep]
                for r in keep[::5]:
                    got = O.py_delfi_single_window(b.rows[c], r["start"], r["stop"], 30, bl[c] if use_bl else [],
                                                   gaps[c] if use_gaps else None)
                    assert got == (r["short"], r["long"], r["num_frags"])
                n += len(keep)
    assert n > 100


def test_cleavage(G, A, bams):
    for sec in SECTIONS:
        b = bams[sec]
        for case in G[sec]["cleavage"]:
            c = case["contig"]
            want = A[case["key"]]
            adj_a, adj_b = max(case["start"] - case["left"], 0), min(case["stop"] + case["right"], dict(zip(b.names, b.lengths))[c])
            got = O.c_cleavage(b.fr[c], adj_a, adj_b, None, None, case["quality_threshold"])[2]
            assert np.allclose(got, want, rtol=1e-12, atol=0), (sec, case)


def test_multi_wps_rows(G, A, bams):
    """multi_wps over edge_sites.bed: the per-base rows of every site window, contigs in chrom.sizes order."""
    from tests.helpers import read_bed
    b = bams["edge"]
    sizes = dict(zip(b.names, b.lengths))
    sites = read_bed(os.path.join(GOLDEN, "edge_sites.bed"))
    pos, val = A["edge_multi_wps_pos"], A["edge_multi_wps_val"]
    assert G["edge"]["multi_wps_contigs"] == ["chrA", "chrB"]
    # frag/_multi_wps.py:240-297: windows mid +- 1000 clipped to the contig; a window is cut at the start of the NEXT
    # LINE's window when that lies on the same contig and begins before it ends (the site file's last line, chrA:50-60,
    # begins at 0 and so empties the window of the line before it); every run of consecutive positions in the golden
    # rows is one of the windows left
    wins = []
    for c, a, z, _ in sites:
        mid = (a + z) // 2
        a, z = max(mid - 1000, 0), min(mid + 1000, sizes[c])
        if wins and wins[-1][0] == c and a < wins[-1][2]:
            wins[-1][2] = a
        wins.append([c, a, z])
    expect = {(a, z): c for c, a, z in wins if z > a}
    assert len(expect) == 4
    cuts = np.flatnonzero(np.diff(pos) != 1) + 1
    k = 0
    for run in np.split(np.arange(len(pos)), cuts):
        a, z = int(pos[run[0]]), int(pos[run[-1]]) + 1
        c = expect.pop((a, z))
        assert np.array_equal(O.c_wps(b.fr[c], a, z, sizes[c]), val[run]), (c, a, z)
        k += len(run)
    assert not expect
    assert k == len(pos)


def test_errors_the_reference_raises(G):
    err = G["errors"]
    for key in ("nocigar_negative_tlen_fetch", "nocigar_negative_tlen_coverage", "nocigar_negative_tlen_wps"):
        assert err[key] == dict(ok=False, error="TypeError", message="unsupported operand type(s) for +: 'NoneType' and 'int'")
    with pytest.raises(TypeError):
        O.bam_rows(NOCIGAR)
    assert err["nocigar_negative_tlen_region_without_it"]["value"] == [["chrE", 1000, 1170, 60, True]]
    assert err["unknown_contig"]["error"] == "ValueError" and err["negative_region_start"]["error"] == "ValueError"
    assert err["start_beyond_stop"]["error"] == "ValueError" and err["bounds_without_contig"]["error"] == "InvalidInputError"
    assert err["region_beyond_contig"] == dict(ok=True, value=0)
