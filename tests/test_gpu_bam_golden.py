"""
GPU: BAM input through the device inflate + device record parser + the kernels, against vectors the REFERENCE'S OWN
code produced in BAM mode (tests/golden/bam.json.gz, bam.npz: oracle/gen_golden_bam.py runs the imported reference
over a BAM stand-in for pysam on its 48-record fixture and on tests/golden/edge.bam).  What the edge file holds:
soft / hard clips, I / D / N / = / X / P / B CIGAR ops, read2 in front of read1, every rejected flag alone, TLEN 0,
TLEN inconsistent with the alignment, CIGAR-less read1 records (TLEN > 0), alignments that consume no reference,
read1 alignments in another window than their fragment's midpoint, read1 poking out of its fragment, unplaced reads.

Two documented departures, both loud (csrc/ftk_bamrule.h, include/ftk.h `ftk_fragstream_skipped`):
  * a fragment with a NEGATIVE start (reference_end + TLEN < 0; contig chrN of the edge file) - the reference keeps it,
    the int32 columns cannot: dropped with a UserWarning.  Section `negative_start` is therefore held to the reference
    wherever those four fragments make no difference, and to the oracle without them where they do;
  * a CIGAR-less read1 with TLEN < 0 - the reference raises TypeError (None + int): so does this.
"""
import os
import warnings

import numpy as np
import pytest

from finaletoolkit_amd import frag, source
from finaletoolkit_amd.exceptions import InvalidInputError
from finaletoolkit_amd.io import AlignmentWrapper
from finaletoolkit_amd.utils import frag_array, frag_generator
from oracle import oracle as O
from tests.helpers import DATA, GOLDEN
from tests.test_oracle_golden_bam import EDGE, FIX, NOCIGAR, bam_golden

pytestmark = pytest.mark.gpu

SIZES = {"12": 133_851_895, "chrA": 400_000, "chrB": 150_000, "chrN": 60_000, "chrZ": 10_000}
PATH = {"fixture": FIX, "edge": EDGE, "negative_start": EDGE}


@pytest.fixture(scope="module")
def G():
    return bam_golden()


@pytest.fixture(scope="module")
def A():
    return np.load(os.path.join(GOLDEN, "bam.npz"))


@pytest.fixture(scope="module")
def chrN():
    """chrN's rows as the reference yields them, and without the fragments whose start is negative."""
    _, _, rows = O.bam_rows(EDGE)
    kept = [r for r in rows["chrN"] if r[0] >= 0]
    assert len(rows["chrN"]) - len(kept) == 4
    return dict(all=rows["chrN"], kept=kept, fr_all=O.frags_from_bam_rows(rows["chrN"])[0], fr=O.frags_from_bam_rows(kept)[0])


@pytest.fixture(autouse=True)
def _quiet():
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        yield


def _rows(it):
    return [[f[0], int(f[1]), int(f[2]), int(f[3]), bool(f[4])] for f in it]


def _drop_neg(rows):
    return [r for r in rows if r[1] >= 0]


def _py(kw):
    return (kw.get("min_length"), kw.get("max_length"), kw.get("intersect_policy", "midpoint"), kw.get("quality_threshold", 30))


@pytest.mark.parametrize("sec", ["fixture", "edge", "negative_start"])
def test_alignment_wrapper_fetch(G, sec):
    """AlignmentWrapper.fetch: whole file, whole contigs and regions (read1 alignments overlapping the region)."""
    path = PATH[sec]
    for q in (0, 30):
        with AlignmentWrapper(path, quality_threshold=q) as aw:
            if sec != "negative_start":
                assert dict(aw.chroms) == G[sec]["chroms"] and aw.is_sam
            for case in G[sec]["fetch"]:
                if case["quality_threshold"] != q:
                    continue
                got = _rows(aw.fetch(case["contig"], case["start"], case["stop"]))
                assert got == _drop_neg(case["fragments"]), (sec, case["contig"], case["start"], case["stop"], q)


@pytest.mark.parametrize("sec", ["fixture", "edge", "negative_start"])
def test_frag_generator_and_frag_array(G, sec):
    path = PATH[sec]
    for case in G[sec]["frag_generator"]:
        got = _rows(frag_generator(path, case["contig"], start=case["start"], stop=case["stop"], **case["kw"]))
        assert got == _drop_neg(case["fragments"]), (sec, case["contig"], case["start"], case["stop"], case["kw"])
    for case in G[sec]["frag_array"]:
        arr = frag_array(path, case["contig"], start=case["start"], stop=case["stop"], **case["kw"])
        got = [[int(r["start"]), int(r["stop"]), bool(r["strand"])] for r in arr]
        assert got == [r for r in case["rows"] if r[0] >= 0], (sec, case)


@pytest.mark.parametrize("sec", ["fixture", "edge"])
def test_single_coverage(G, sec):
    path = PATH[sec]
    for case in G[sec]["single_coverage"]:
        if case["contig"] is None and sec == "edge":
            continue  # the whole edge file holds chrN's negative starts: test_negative_starts_are_dropped_loudly
        r = frag.single_coverage(path, case["contig"], case["start"], case["stop"], **case["kw"])
        assert r.coverage == case["coverage"], (sec, case)


def test_coverage_driver(G, tmp_path):
    iv = os.path.join(GOLDEN, "edge_intervals.bed")
    for key, kw in (("coverage_default", {}), ("coverage_any_q0", dict(intersect_policy="any", quality_threshold=0)),
                    ("coverage_len_120_180", dict(min_length=120, max_length=180))):
        got = frag.coverage(EDGE, iv, None, **kw)
        assert [list(r) for r in got] == G["edge"][key], key
    fx = frag.coverage(FIX, os.path.join(DATA, "intervals.bed"), None, normalize=False)
    assert [list(r) for r in fx] == G["fixture"]["coverage_raw"]
    fx = frag.coverage(FIX, os.path.join(DATA, "intervals.bed"), None, normalize=True)
    assert [list(r) for r in fx] == G["fixture"]["coverage_norm"]


@pytest.mark.parametrize("sec", ["fixture", "edge"])
def test_wps(G, A, sec):
    path = PATH[sec]
    for case in G[sec]["wps"]:
        r = frag.wps(path, case["contig"], case["start"], case["stop"], SIZES[case["contig"]], window_size=case["window_size"],
                     min_length=case["min_length"], max_length=case["max_length"], quality_threshold=case["quality_threshold"])
        assert np.array_equal(r["wps"].astype(np.int64), A[case["key"]]), (sec, case)
        assert r["start"][0] == case["start"] and len(r) == case["stop"] - case["start"]


@pytest.mark.parametrize("sec", ["fixture", "edge"])
def test_lengths(G, A, sec):
    path = PATH[sec]
    for case in G[sec]["frag_length"]:
        got = frag.frag_length(path, contig=case["contig"], start=case["start"], stop=case["stop"], **case["kw"])
        assert np.asarray(got).tolist() == A[case["key"]].tolist(), (sec, case)
    for case in G[sec]["frag_length_bins"]:
        if "contig" not in case["kw"] and sec == "edge":
            continue  # genome-wide over the edge file: chrN's negative starts (below)
        bins, counts = frag.frag_length_bins(path, **case["kw"])
        assert np.asarray(bins).tolist() == case["bins"] and list(map(int, counts)) == case["counts"], (sec, case["kw"])


def _stats_equal(got, want):
    assert list(got[:4]) == list(want[:4])
    assert got[5] == want[5] and list(got[7:10]) == list(want[7:10])
    assert got[4] == pytest.approx(want[4], rel=1e-12)
    assert got[6] == pytest.approx(want[6], rel=1e-9)      # stdev: summation order differs
    assert got[10] == pytest.approx(want[10], rel=1e-12)


def test_length_intervals(G):
    iv = os.path.join(GOLDEN, "edge_intervals.bed")
    for got, want in zip(frag.frag_length_intervals(EDGE, iv), G["edge"]["frag_length_intervals"], strict=True):
        _stats_equal(got, want)
    got = frag.frag_length_intervals(EDGE, iv, min_length=50, max_length=600, intersect_policy="any", quality_threshold=0,
                                     short_reads=167)
    for g, w in zip(got, G["edge"]["frag_length_intervals_any_q0"], strict=True):
        _stats_equal(g, w)
    for g, w in zip(frag.frag_length_intervals(FIX, os.path.join(DATA, "intervals.bed")), G["fixture"]["frag_length_intervals"],
                    strict=True):
        _stats_equal(g, w)


@pytest.mark.parametrize("sec", ["fixture", "edge"])
def test_cleavage(G, A, sec):
    path = PATH[sec]
    for case in G[sec]["cleavage"]:
        r = frag.cleavage_profile(path, SIZES[case["contig"]], case["contig"], case["start"], case["stop"], left=case["left"],
                                  right=case["right"], quality_threshold=case["quality_threshold"])
        assert np.allclose(r["proportion"], A[case["key"]], rtol=1e-12, atol=0), (sec, case)


def test_delfi_windows_through_the_c_abi(G):
    """ftk_delfi_counts on the BAM contigs (read1 fetch mode) = the reference's _delfi_single_window rows."""
    src = source.open_source(EDGE)
    eng = source.get_engine()
    bl = {c: sorted(zip(*v)) for c, v in G["edge"]["blacklist"].items()}
    gaps = {c: (g["centromere"][0], g["centromere"][1], [tuple(t) for t in g["telomeres"]]) for c, g in G["edge"]["gaps"].items()}
    n = 0
    for use_gaps in (True, False):
        for use_bl in (True, False):
            for c in ("chrA", "chrB"):
                rows = [r for r in G["edge"]["delfi_windows"] if r["gaps"] == use_gaps and r["blacklist"] == use_bl
                        and r["contig"] == c and r["arm"] != "NOARM"]
                if not rows:
                    continue
                ws = np.array([r["start"] for r in rows], np.int32)
                we = np.array([r["stop"] for r in rows], np.int32)
                bs = np.array([x[0] for x in bl[c]], np.int32) if use_bl else None
                be = np.array([x[1] for x in bl[c]], np.int32) if use_bl else None
                sh, lg, nf = eng.delfi_counts(src.require(c), ws, we, 30, bs, be, gaps[c] if use_gaps else None)
                assert sh.tolist() == [r["short"] for r in rows] and lg.tolist() == [r["long"] for r in rows], (c, use_gaps, use_bl)
                assert nf.tolist() == [r["num_frags"] for r in rows]
                n += len(rows)
    assert n > 100


def test_multi_wps(G, A, tmp_path):
    import gzip
    out = str(tmp_path / "mw.bed.gz")
    frag.multi_wps(EDGE, os.path.join(GOLDEN, "edge_sites.bed"), os.path.join(GOLDEN, "edge.chrom.sizes"), out, interval_size=2000)
    rows = [ln.split("\t") for ln in gzip.open(out, "rt").read().splitlines()]
    assert np.array_equal(np.array([int(r[1]) for r in rows]), A["edge_multi_wps_pos"])
    assert np.array_equal(np.array([int(r[3]) for r in rows]), A["edge_multi_wps_val"])
    assert sorted({r[0] for r in rows}) == G["edge"]["multi_wps_contigs"]


def test_negative_starts_are_dropped_loudly(G, A, chrN):
    """Section `negative_start` (contig chrN): equal to the reference wherever its four negative-start fragments make
    no difference, equal to the oracle WITHOUT them where they do, and a UserWarning says so."""
    source.close_all()
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        assert frag.single_coverage(EDGE, "chrN", 0, None, quality_threshold=0).coverage == \
            O.py_single_coverage(chrN["kept"], 0, None, None, None, "midpoint", 0)
    assert any("starts before position 0" in str(w.message) and issubclass(w.category, UserWarning) for w in seen)
    sec = G["negative_start"]
    same = diff = 0
    for case in sec["single_coverage"]:
        kw = case["kw"]
        want = O.py_single_coverage(chrN["kept"], case["start"], case["stop"], *_py(kw))
        got = frag.single_coverage(EDGE, "chrN", case["start"], case["stop"], **kw).coverage
        assert got == want, case
        same += want == case["coverage"]
        diff += want != case["coverage"]
    assert same > 30 and diff >= 3   # (the negative fragments count under the `any` policy and in the whole contig)
    for case in sec["wps"]:
        want = O.c_wps(chrN["fr"], case["start"], case["stop"], SIZES["chrN"], case["window_size"], case["min_length"],
                       case["max_length"], case["quality_threshold"])
        r = frag.wps(EDGE, "chrN", case["start"], case["stop"], SIZES["chrN"], window_size=case["window_size"],
                     min_length=case["min_length"], max_length=case["max_length"], quality_threshold=case["quality_threshold"])
        assert np.array_equal(r["wps"].astype(np.int64), want), case
        if case["start"] >= 5_000:
            assert np.array_equal(want, A[case["key"]])
    assert not np.array_equal(O.c_wps(chrN["fr"], 0, 800, SIZES["chrN"], 120, 20, 500, 0), A["neg_wps_1"])
    for case in sec["frag_length"]:
        kw = case["kw"]
        got = frag.frag_length("%s" % EDGE, contig="chrN", start=case["start"], stop=case["stop"], **kw)
        want = [r[1] - r[0] for r in O.py_frag_generator(chrN["kept"], case["start"], case["stop"], 0, 1_000_000_000,
                                                         kw.get("intersect_policy", "midpoint"), kw.get("quality_threshold", 30))]
        assert np.asarray(got).tolist() == want
    # the whole file's total (normalisation) = the reference's minus the negative-start fragments that pass midpoint >= 0
    whole = [c for c in G["edge"]["single_coverage"] if c["contig"] is None][0]
    got = frag.single_coverage(EDGE, None, 0, None, quality_threshold=0).coverage
    lost = O.py_single_coverage(chrN["all"], 0, None, None, None, "midpoint", 0) - O.py_single_coverage(chrN["kept"], 0, None, None, None, "midpoint", 0)
    assert got == whole["coverage"] - lost


def test_errors_like_the_reference(G):
    err = G["errors"]
    source.close_all()
    for call in (lambda: list(AlignmentWrapper(NOCIGAR, quality_threshold=0).fetch("chrE")),
                 lambda: frag.single_coverage(NOCIGAR, "chrE", 0, None),
                 lambda: frag.wps(NOCIGAR, "chrE", 1_900, 2_100, 20_000)):
        with pytest.raises(TypeError, match="NoneType"):
            call()
        source.close_all()
    assert err["nocigar_negative_tlen_fetch"]["error"] == "TypeError"
    for key, call in (("unknown_contig", lambda: frag.single_coverage(EDGE, "chrQ", 0, 100)),
                      ("negative_region_start", lambda: frag.single_coverage(EDGE, "chrA", -5, 100)),
                      ("start_beyond_stop", lambda: frag.single_coverage(EDGE, "chrA", 500, 100))):
        assert err[key]["error"] == "ValueError"
        with pytest.raises(ValueError):
            call()
    assert err["bounds_without_contig"]["error"] == "InvalidInputError"
    with pytest.raises(InvalidInputError):
        frag.single_coverage(EDGE, None, 5, 100)
    assert frag.single_coverage(EDGE, "chrZ", 50_000, 60_000).coverage == err["region_beyond_contig"]["value"] == 0


@pytest.mark.parametrize("mode", ["device", "host_walk", "host_all"])
def test_every_decoder_path_yields_the_reference_rows(G, mode, monkeypatch):
    """The three BAM paths (records parsed on the device / device inflate + host walk / all host) hand out the same,
    reference-pinned rows for the edge file."""
    if mode == "host_walk":
        monkeypatch.setenv("FTK_DEVICE_BAM_PARSE", "0")
    if mode == "host_all":
        monkeypatch.setenv("FTK_DEVICE_INFLATE", "0")
        monkeypatch.setenv("FTK_DEVICE_BAM_PARSE", "0")
    source.close_all()
    for case in G["edge"]["fetch"]:
        if case["start"] is None and case["contig"] is not None and case["quality_threshold"] == 0:
            got = _rows(frag_generator(EDGE, case["contig"], quality_threshold=0))
            assert got == case["fragments"], (mode, case["contig"])
    source.close_all()
