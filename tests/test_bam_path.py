"""BAM input: CPU test of the decoder's flag / TLEN / read1 rules on a synthetic BAM, and a
GPU test of the whole file path in BAM_READ1 fetch mode against the oracle."""
import numpy as np
import pytest

from finaletoolkit_amd import synth
from tests.helpers import write_synthetic_bam
from tests.test_abi import _decode

CONTIGS = [("chr1", 400_000), ("chrEmpty", 1000), ("chr2", 250_000)]


def _make(tmp_path, depth=6.0):
    frags = {}
    for i, (c, n) in enumerate(CONTIGS):
        if c == "chrEmpty":
            continue
        s, e, q, st = synth.synth_contig(n, depth=depth, seed=900 + i)
        frags[c] = (s, e, q, st)
    path = str(tmp_path / "syn.bam")
    return path, write_synthetic_bam(path, CONTIGS, frags)


def _order(path, contig):
    """File-order ranks of a BAM contig (ftk_fragtable_order)."""
    import ctypes as C
    from finaletoolkit_amd import _lib as L
    lib = L.load()
    t = C.c_void_p()
    assert lib.ftk_bam_decode(path.encode(), contig.encode(), 2, C.byref(t)) == 0
    try:
        p = C.c_void_p()
        assert lib.ftk_fragtable_order(t, 0, C.byref(p)) == 0 and p.value
        n = lib.ftk_fragtable_contig_rows(t, 0)
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_int32)), (n,)).astype(np.int64)
    finally:
        lib.ftk_fragtable_free(t)


def test_bam_decoder_rules(tmp_path):
    path, expected = _make(tmp_path, depth=3.0)
    got = _decode(path, bam=True, threads=3)
    assert got["chrEmpty"][0] == 0 and got["chrEmpty"][2] == 1000
    for c in ("chr1", "chr2"):
        rows, cols, length = got[c]
        want = np.array(expected[c], dtype=np.int64)
        assert rows == len(want) and length == dict(CONTIGS)[c]
        for k in range(6):
            assert np.array_equal(np.asarray(cols[k], dtype=np.int64), want[:, k]), (c, k)
        assert np.array_equal(_order(path, c), want[:, 6]), c
        assert not np.array_equal(want[:, 6], np.arange(len(want)))  # start order != file order in this file
    only = _decode(path, bam=True, contig="chr2")
    assert [k for k in only if not k.startswith("__")] == ["chr2"]


@pytest.mark.gpu
def test_bam_file_path_read1_semantics(tmp_path):
    from finaletoolkit_amd import frag
    from oracle import oracle as O
    path, expected = _make(tmp_path)
    for c, size in (("chr1", 400_000), ("chr2", 250_000)):
        w = np.array(expected[c], dtype=np.int64)
        fr = O.Frags(w[:, 0], w[:, 1], w[:, 2], w[:, 3], w[:, 4], w[:, 5])
        ws, we = synth.tiling_windows(size, 5_000)
        for policy in ("midpoint", "any"):
            want = O.c_window_counts(fr, ws, we, mapq_min=30, policy=policy)
            got = [frag.single_coverage(path, c, int(a), int(b), intersect_policy=policy).coverage
                   for a, b in zip(ws[:40], we[:40])]
            assert got == want[:40].tolist(), (c, policy)
        # whole-contig fetch (no window): every read1 record of the contig
        assert frag.single_coverage(path, c, 0, None, quality_threshold=0).coverage == len(w)
        r = frag.wps(path, c, 100_000, 104_000, size)
        assert np.array_equal(r["wps"], O.c_wps(fr, 100_000, 104_000, size, 120, 120, 180, 30))
    # the htslib quirk the reference documents (tests/test_delfi.py:135-139): a fragment whose midpoint is
    # in the window but whose read1 lies outside is NOT counted for BAM input
    w = np.array(expected["chr1"], dtype=np.int64)
    mid = (w[:, 0] + w[:, 1]) // 2
    rev = np.nonzero((w[:, 3] == 0) & (w[:, 2] >= 30) & (w[:, 1] - w[:, 0] > 210))[0]
    i = int(rev[len(rev) // 2])
    a, b = int(mid[i]) - 5, int(mid[i]) + 1  # midpoint inside, read1 (at the fragment end) outside
    assert w[i, 4] >= b
    tab_like = int(((mid >= a) & (mid < b) & (w[:, 2] >= 30)).sum())
    got = frag.single_coverage(path, "chr1", a, b).coverage
    assert got < tab_like
    assert frag.single_coverage(path, "chrEmpty", 0, None).coverage == 0
    # frag_length / frag_generator / frag_array hand rows back in pysam's order (read1 position in the file),
    # not in the start order the kernels work in
    from finaletoolkit_amd.utils import frag_array, frag_generator
    w = np.array(expected["chr2"], dtype=np.int64)
    by_file = w[np.argsort(w[:, 6])]
    keep = by_file[(by_file[:, 2] >= 20)]
    assert np.array_equal(frag.frag_length(path, contig="chr2", quality_threshold=20), (keep[:, 1] - keep[:, 0]))
    gen = list(frag_generator(path, "chr2", 20))
    assert [(g[1], g[2]) for g in gen[:500]] == [(int(a), int(b)) for a, b in keep[:500, :2]]
    a, b = 100_000, 140_000
    sel = by_file[(by_file[:, 2] >= 30) & (by_file[:, 4] < b) & (by_file[:, 5] > a)
                  & (((by_file[:, 0] + by_file[:, 1]) // 2) >= a) & (((by_file[:, 0] + by_file[:, 1]) // 2) < b)]
    arr = frag_array(path, "chr2", 30, a, b)
    assert np.array_equal(arr["start"], sel[:, 0]) and np.array_equal(arr["stop"], sel[:, 1])


@pytest.mark.gpu
def test_lazy_sources_decode_only_what_is_asked_for(tmp_path, monkeypatch):
    """Indexed inputs are opened lazily: an interval query reads the interval's records through the index (or decodes one
    contig when the interval is open-ended), whole-file operations load the rest, and the results equal the one-pass decode (FTK_LAZY_SOURCE=0)."""
    from finaletoolkit_amd import bgzf, frag, source
    path, expected = _make(tmp_path)                      # BAM + real (minimal) BAI
    rows = []
    for k, size in enumerate((600_000, 250_000, 90_000)):
        s, e, q, st = synth.synth_contig(size, depth=10.0, seed=70 + k)
        rows.append((f"t{k}", s, e, q, st))
    text = str(tmp_path / "lazy.frag.gz")
    bgzf.write_frag_gz(text, rows, level=1, with_index=True)

    def run():
        out = [frag.single_coverage(path, "chr2", 10_000, 200_000).coverage,
               frag.single_coverage(text, "t1", 0, None, quality_threshold=0).coverage]
        loaded = (set(source.open_source(path).loaded), set(source.open_source(text).loaded))
        out.append(frag.single_coverage(path, None, 0, None, quality_threshold=0).coverage)   # whole file
        out.append(frag.single_coverage(text, None, 0, None, quality_threshold=0).coverage)
        out.append(len(frag.frag_length(text, contig="t2", quality_threshold=10)))
        with pytest.raises(ValueError):
            frag.single_coverage(text, "nope", 0, 10)
        return out, loaded

    source.close_all()
    lazy, loaded = run()
    # the BAM interval came through the BAI as a region of chr2 (no contig decoded); the open-ended text query loaded t1
    assert loaded == (set(), {"t1"}) and source.REGION_READS[-1][1:] == ("chr2", 9_999, 200_001)  # (one base of padding)
    assert set(source.open_source(path).loaded) == {"chr1", "chrEmpty", "chr2"}
    assert set(source.open_source(text).loaded) == {"t0", "t1", "t2"}
    assert lazy[1] == len(rows[1][1]) and lazy[3] == sum(len(r[1]) for r in rows)
    source.close_all()
    monkeypatch.setenv("FTK_LAZY_SOURCE", "0")
    eager, loaded = run()
    assert loaded[0] == {"chr1", "chrEmpty", "chr2"} and loaded[1] == {"t0", "t1", "t2"}
    assert eager == lazy
    source.close_all()


@pytest.mark.gpu
def test_bam_wps_interval_calls_through_the_index_equal_the_whole_contig(tmp_path):
    """frag.wps on an interval of a BAM contig that is not resident reads a REGION through the BAI.  The reference's
    fetch window for WPS is [start - max_length, stop + max_length) (frag/_wps.py:156-157) and a BAM query returns
    read1 ALIGNMENTS overlapping it, so with 50 bp reads a read1 lying up to 180 bp outside the interval still brings its
    fragment in: the region must be padded by max_length, not by the WPS window.  Intervals end within 180 bp of a
    16 kb linear-index boundary (where a narrower pad would lose records); every answer must equal the call on the
    resident contig and the oracle."""
    from finaletoolkit_amd import frag, source
    from oracle import oracle as O
    size = 1_500_000
    path = str(tmp_path / "r50.bam")
    exp = synth.write_paired_bam(path, "w", size, 120.0, 77, read_len=50)
    fr = O.Frags(exp["s"], exp["e"], exp["q"], exp["st"], exp["r1s"], exp["r1e"])
    cases = []
    for k in (5, 23, 41, 60, 77):
        for d in (0, 35, 120, 150, 179):
            cases.append((k * 16384 - d - 3000, k * 16384 - d))       # the interval ENDS d bases before a boundary
            cases.append((k * 16384 + d, k * 16384 + d + 2500))       # ... or STARTS d bases behind one
    got_region = []
    for a, b in cases:
        source.close_all()                                            # a fresh source: its first interval call -> region
        del source.REGION_READS[:]
        got_region.append(frag.wps(path, "w", a, b, size)["wps"])
        assert len(source.REGION_READS) == 1 and source.REGION_READS[0][2:] == (a - 181, b + 181)
        assert not source.open_source(path).loaded
    source.close_all()
    source.open_source(path).require("w")
    differ_from_slice = 0
    for (a, b), r in zip(cases, got_region):
        whole = frag.wps(path, "w", a, b, size)["wps"]
        want = O.c_wps(fr, a, b, size, 120, 120, 180, 30)
        assert np.array_equal(whole, want), (a, b)
        assert np.array_equal(r, want), (a, b)
        differ_from_slice += int(r.any())
    assert differ_from_slice == len(cases)
    source.close_all()


def _golden_rows(contig):
    """Whole-contig rows of tests/golden/edge.bam as the imported reference's AlignmentWrapper yielded them
    (oracle/gen_golden_bam.py), file order; fragments with a negative start removed (the int32 columns cannot
    hold them: counted by ftk_*_skipped instead)."""
    from tests.test_oracle_golden_bam import bam_golden
    G = bam_golden()
    for sec in ("edge", "negative_start"):
        for case in G[sec]["fetch"]:
            if case["contig"] == contig and case["start"] is None and case["stop"] is None and case["quality_threshold"] == 0:
                return [r for r in case["fragments"] if r[1] >= 0], sum(r[1] < 0 for r in case["fragments"])
    raise KeyError(contig)


def test_host_decoders_hold_the_reference_generated_rows():
    """ftk_bam_decode and the host stream on the edge BAM (multi-op CIGARs, CIGAR-less records, alignments that
    consume no reference, every rejected flag): the rows the REFERENCE yields, in its order, the read1 spans
    htslib's iterator would test; what it cannot hold is counted, not silently dropped."""
    import ctypes as C
    import os
    from finaletoolkit_amd import _lib as L
    from oracle import oracle as O
    from tests.helpers import GOLDEN
    from tests.test_stream_decoder import _stream
    path = os.path.join(GOLDEN, "edge.bam")
    _, _, rows = O.bam_rows(path)  # (held to the same goldens by tests/test_oracle_golden_bam.py)
    whole = _decode(path, bam=True, threads=3)
    streamed, order, refs = _stream(path, bam=True, threads=2)
    assert [r[0] for r in refs] == ["chrA", "chrB", "chrN", "chrZ"] and order == ["chrA", "chrB", "chrN"]
    n_neg = 0
    for c in ("chrA", "chrB", "chrN"):
        want, neg = _golden_rows(c)
        n_neg += neg
        r1 = {}
        for r in rows[c]:
            r1.setdefault(r[:4], []).append(r[4:6])
        for got in (whole[c], streamed[c]):
            n, cols, _ = got
            assert n == len(want), c
            rank = _order(path, c)
            by_file = np.argsort(rank, kind="stable")
            s, e, q, st, a, b = [np.asarray(x)[by_file].tolist() for x in cols]
            assert [[c, s[i], e[i], q[i], bool(st[i])] for i in range(n)] == want, c
            for i in range(n):  # the read1 span of every row: [pos, bam_endpos)
                assert (a[i], b[i]) in r1[(s[i], e[i], q[i], st[i])], (c, i)
            assert np.all(np.diff(np.asarray(cols[0], np.int64)) >= 0)
    assert whole["chrZ"][0] == 0 and n_neg == 4
    lib = L.load()
    out = (C.c_int64 * 2)()
    t = C.c_void_p()
    assert lib.ftk_bam_decode(path.encode(), None, 2, C.byref(t)) == 0
    assert lib.ftk_fragtable_skipped(t, C.byref(out)) == 0 and list(out) == [4, 0]
    lib.ftk_fragtable_free(t)
    t = C.c_void_p()
    assert lib.ftk_bam_decode(os.path.join(GOLDEN, "edge_nocigar.bam").encode(), None, 1, C.byref(t)) == 0
    assert lib.ftk_fragtable_skipped(t, C.byref(out)) == 0 and list(out) == [0, 1]   # the reference raises TypeError there
    assert lib.ftk_fragtable_contig_rows(t, 0) == 1
    lib.ftk_fragtable_free(t)
    s = C.c_void_p()
    assert lib.ftk_fragstream_open(path.encode(), b"chrN", 1, 2, 1, C.byref(s)) == 0
    t = C.c_void_p()
    assert lib.ftk_fragstream_next(s, C.byref(t)) == 0 and t.value
    assert lib.ftk_fragstream_skipped(s, C.byref(out)) == 0 and list(out) == [4, 0]
    lib.ftk_fragtable_free(t)
    lib.ftk_fragstream_close(s)
