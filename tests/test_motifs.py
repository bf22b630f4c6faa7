"""End-motif / breakpoint-motif features: the oracle against the reference's recorded outputs
(CPU) and the HIP path against both, on FASTA and 2bit images of the same genome (GPU).
Counts are exact; frequencies are counts / total in float64, compared exactly."""
import hashlib
import json
import os
import warnings

import numpy as np
import pytest

from oracle import oracle as O
from tests import helpers as H

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
FRAGS = os.path.join(GOLD, "synth.frag.gz")
BED = os.path.join(GOLD, "motif_intervals.bed")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "motifs.npz")), json.load(open(os.path.join(GOLD, "motifs.json")))


@pytest.fixture(scope="module")
def genome():
    return H.read_fasta_gz(os.path.join(GOLD, "synth_ref.fa.gz"))


@pytest.fixture(scope="module")
def rows():
    cols = H.read_frag_gz(FRAGS)
    return {c: list(zip(*(a.tolist() for a in v))) for c, v in cols.items()}


@pytest.fixture(scope="module")
def refs(genome, tmp_path_factory, gold):
    """FASTA and 2bit files of the synthetic genome, plus the variants with chrB cut short."""
    d = tmp_path_factory.mktemp("ref")
    short = {"chrA": genome["chrA"], "chrB": genome["chrB"][:gold[1]["short_b"]]}
    out = {}
    for tag, seqs in (("full", genome), ("short", short)):
        H.write_fasta(d / f"{tag}.fa", seqs)
        H.write_2bit(d / f"{tag}.2bit", seqs)
        out[tag] = (str(d / f"{tag}.fa"), str(d / f"{tag}.2bit"))
    H.write_fasta(d / "noindex.fa", genome, width=73, fai=False)
    out["noindex"] = str(d / "noindex.fa")
    return out


def bed_rows(path):
    return [(f[0], int(f[1]), int(f[2])) for f in (l.split() for l in open(path))]


def kind_of(key):
    return "breakpoint" if "bp_" in key else "end"


def oracle_kw(kw):
    return dict(k=kw.get("k"), both_strands=kw.get("both_strands", True),
                negative_strand=kw.get("negative_strand", False), quality_threshold=kw.get("quality_threshold", 30))


# ---------------------------------------------------------------------------------- CPU
def test_oracle_matches_reference_intervals(gold, genome, rows):
    A, J = gold
    for key in ("iv_end_k4_both", "iv_end_k2_neg_q0", "iv_bp_k4_both", "iv_bp_k6_fwd"):
        kw = oracle_kw(J[key]["kw"])
        for i, (c, s, e) in enumerate(bed_rows(BED)):
            got = O.py_region_motifs(rows[c], genome[c], s, e, kind=kind_of(key), **kw)
            assert np.array_equal(got, A[key][i]), (key, i)
        for i, (c, s, e, _) in enumerate(J["regions"]):
            got = O.py_region_motifs(rows.get(c, []), genome.get(c), s, e, kind=kind_of(key), **kw)
            assert np.array_equal(got, A[key + "_regions"][i]), (key, "regions", i)


def test_oracle_matches_reference_off_contig(gold, genome, rows):
    A, J = gold
    short = genome["chrB"][:J["short_b"]]
    r = rows["chrB"]
    assert J["short_end_both"] == "RuntimeError"
    with pytest.raises(RuntimeError):
        O.py_region_motifs(r, short, 140_000, 150_000, 4, quality_threshold=20)
    assert np.array_equal(O.py_region_motifs(r, short, 140_000, 150_000, 4, both_strands=False), A["short_end_k4_fwd"])
    assert np.array_equal(O.py_region_motifs(r, short, 140_000, 150_000, 4, both_strands=False, negative_strand=True),
                          A["short_end_k4_neg"])
    assert np.array_equal(O.py_region_motifs(r, short, 140_000, 150_000, 4, kind="breakpoint", quality_threshold=30),
                          A["short_bp_k4_both"])
    assert np.array_equal(O.py_region_motifs(r, short, 140_000, 150_000, 4, kind="breakpoint", both_strands=False,
                                             negative_strand=True, quality_threshold=30), A["short_bp_k4_neg"])


def test_oracle_matches_reference_genome_wide(gold, genome, rows):
    from finaletoolkit_amd.frag._motif_common import genome_windows
    A, J = gold
    wins = genome_windows({c: len(s) for c, s in genome.items()})
    assert wins == [("chrA", 0, 400_000), ("chrB", 0, 150_000)]
    for key in ("end_k4_both", "end_k3_fwd", "bp_k4_fwd"):
        kw = oracle_kw(J[key]["kw"])
        tot = sum(O.py_region_motifs(rows[c], genome[c], s, e, kind=kind_of(key), **kw) for c, s, e in wins)
        assert np.array_equal(tot / tot.sum(), A[key]), key


def test_genome_windows_tiling():
    from finaletoolkit_amd.frag._motif_common import genome_windows
    w = genome_windows({"a": 2_500_000, "b": 3_000_000, "c": 10})
    assert w == [("a", 0, 1_000_000), ("a", 1_000_000, 2_000_000), ("a", 2_000_000, 2_500_000),
                 ("b", 0, 1_000_000), ("b", 1_000_000, 2_000_000), ("b", 3_000_000, 3_000_000), ("c", 0, 10)] or \
        w[5] == ("b", 2_000_000, 3_000_000)


def test_containers_roundtrip(tmp_path):
    from finaletoolkit_amd.frag import EndMotifFreqs, EndMotifsIntervals
    from finaletoolkit_amd.frag._motif_common import gen_kmers, normalized_shannon_mds
    kmers = gen_kmers(2)
    assert kmers[:5] == ["AA", "AC", "AG", "AT", "CA"] and len(kmers) == 16
    f = np.arange(16, dtype=float)
    fr = EndMotifFreqs(zip(kmers, f / f.sum()), 2, 30)
    fr.to_tsv(tmp_path / "f.tsv")
    back = EndMotifFreqs.from_file(tmp_path / "f.tsv", 0)
    assert back.freq_dict == pytest.approx(fr.freq_dict) and back.k == 2
    assert fr.motif_diversity_score() == pytest.approx(normalized_shannon_mds(f / f.sum(), 2))
    assert normalized_shannon_mds(np.full(16, 1 / 16), 2) == pytest.approx(1.0)
    with pytest.raises(ValueError):
        EndMotifFreqs([("AAA", 1.0)], 2)
    iv = EndMotifsIntervals([(("c", 0, 10, "n"), dict(zip(kmers, range(16)))),
                             (("c", 10, 20, "."), dict(zip(kmers, [0] * 16)))], 2, 30)
    iv.to_tsv(tmp_path / "iv.csv", calc_freq=False, sep=",")
    back = EndMotifsIntervals.from_file(str(tmp_path / "iv.csv"), 30)
    assert back.k == 2 and back.total_counts == [120.0, 0.0] and back.intervals[0][1]["TG"] == 14.0
    assert "TT\n" in back.intervals[0][1]   # as in the reference: the header line's newline stays on the last k-mer
    iv.to_tsv(tmp_path / "iv.tsv")
    lines = open(tmp_path / "iv.tsv").read().splitlines()
    assert lines[1].split("\t")[4:7] == ["120", "0.000000", "0.008333"] and lines[2].split("\t")[5] == "NaN"
    mds = iv.motif_diversity_score()
    assert 0 < mds[0][1] < 1 and np.isnan(mds[1][1])
    assert np.isnan(iv.motif_diversity_score(miller_madow=True)[1][1])
    iv.mds_bed(tmp_path / "mds.bed")
    assert open(tmp_path / "mds.bed").read().splitlines()[0].startswith("c\t0\t10\tn\t0.")
    iv.to_bed("AC", tmp_path / "ac.bed")
    assert open(tmp_path / "ac.bed").read().splitlines()[0] == "c\t0\t10\tn\t0.008333"


def test_alias_rules():
    from finaletoolkit_amd.frag._motif_common import resolve_motif_aliases
    with pytest.warns(DeprecationWarning):
        assert resolve_motif_aliases(None, None, 40, 300) == (40, 300)
    with pytest.warns(DeprecationWarning), pytest.raises(ValueError):
        resolve_motif_aliases(50, None, 40, None)
    assert resolve_motif_aliases(50, None, None, None) == (50, None)


# ---------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1])  # FASTA image, 2bit image
def test_gpu_genome_wide_matches_reference(engine, gold, refs, which, tmp_path):
    from finaletoolkit_amd import frag
    A, J = gold
    ref = refs["full"][which]
    fns = {"end": frag.end_motifs, "breakpoint": frag.breakpoint_motifs}
    for key in ("end_k4_both", "end_k3_fwd", "end_k5_neg", "bp_k6_both", "bp_k4_fwd", "bp_k2_neg"):
        r = fns[kind_of(key)](FRAGS, ref, **J[key]["kw"])
        assert np.array_equal(np.array(r.frequencies()), A[key]), key
        assert r.motif_diversity_score() == pytest.approx(J[key]["mds"], rel=1e-12)
    out = str(tmp_path / "freqs.tsv")
    frag.end_motifs(FRAGS, ref, k=4, output_file=out)
    assert hashlib.sha256(open(out, "rb").read()).hexdigest() == J["end_k4_both"]["tsv_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1])
def test_gpu_intervals_match_reference(engine, gold, refs, which, tmp_path):
    from finaletoolkit_amd import frag
    A, J = gold
    ref = refs["full"][which]
    fns = {"end": frag.interval_end_motifs, "breakpoint": frag.interval_breakpoint_motifs}
    regions = [tuple(r) for r in J["regions"]]
    for key in ("iv_end_k4_both", "iv_end_k2_neg_q0", "iv_bp_k4_both", "iv_bp_k6_fwd"):
        r = fns[kind_of(key)](FRAGS, ref, BED, **J[key]["kw"])
        got = np.array([list(f.values()) for _, f in r.intervals], np.int64)
        assert np.array_equal(got, A[key]), key
        assert [iv[3] for iv, _ in r.intervals][:2] == ["t0", "t25000"]
        r2 = fns[kind_of(key)](FRAGS, ref, regions, **J[key]["kw"])
        got = np.array([list(f.values()) for _, f in r2.intervals], np.int64)
        assert np.array_equal(got, A[key + "_regions"]), key
        if key == "iv_end_k4_both":
            out = str(tmp_path / "iv.tsv")
            r.to_tsv(out)
            assert hashlib.sha256(open(out, "rb").read()).hexdigest() == J[key]["tsv_sha256"]
            r.to_tsv(out, calc_freq=False, sep=",")
            assert hashlib.sha256(open(out, "rb").read()).hexdigest() == J[key]["csv_counts_sha256"]
            np.testing.assert_allclose([m for _, m in r.motif_diversity_score()], A["iv_end_k4_both_mds"],
                                       rtol=1e-12, equal_nan=True)
            np.testing.assert_allclose([m for _, m in r.motif_diversity_score(True)], A["iv_end_k4_both_mds_mm"],
                                       rtol=1e-12, equal_nan=True)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        r = frag.interval_breakpoint_motifs(FRAGS, ref, regions, k=5)
    assert np.array_equal(np.array([list(f.values()) for _, f in r.intervals]), A["iv_bp_k5_odd_regions"])


@pytest.mark.gpu
@pytest.mark.parametrize("which", [0, 1])
def test_gpu_off_contig_rules(engine, gold, refs, which):
    from finaletoolkit_amd import frag
    A, J = gold
    ref = refs["short"][which]
    tail = ("chrB", 140_000, 150_000)
    with pytest.raises(RuntimeError):
        frag.region_end_motifs(FRAGS, *tail, ref, k=4)
    with pytest.raises(ValueError):
        frag.region_end_motifs(FRAGS, *tail, ref, k=4, negative_strand=True)
    val = lambda d: np.array(list(d.values()), np.int64)
    assert np.array_equal(val(frag.region_end_motifs(FRAGS, *tail, ref, k=4, both_strands=False)), A["short_end_k4_fwd"])
    assert np.array_equal(val(frag.region_end_motifs(FRAGS, *tail, ref, k=4, both_strands=False,
                                                     negative_strand=True)), A["short_end_k4_neg"])
    assert np.array_equal(val(frag.region_breakpoint_motifs(FRAGS, *tail, ref, k=4)), A["short_bp_k4_both"])
    assert np.array_equal(val(frag.region_breakpoint_motifs(FRAGS, *tail, ref, k=4, both_strands=False,
                                                            negative_strand=True)), A["short_bp_k4_neg"])


@pytest.mark.gpu
def test_gpu_unindexed_fasta_other_line_width(engine, gold, refs):
    from finaletoolkit_amd import frag
    A, J = gold
    r = frag.end_motifs(FRAGS, refs["noindex"], k=4)
    assert np.array_equal(np.array(r.frequencies()), A["end_k4_both"])


@pytest.mark.gpu
@pytest.mark.parametrize("k,kind", [(1, "end"), (6, "end"), (7, "end"), (6, "breakpoint")])
def test_gpu_large_random_vs_oracle(engine, k, kind, tmp_path):
    """3e5 fragments on a 2 Mb contig, windows from 1 kb to the whole contig (chunked path included)."""
    rng = np.random.default_rng(100 + k)
    L = 2_000_000
    seq = np.frombuffer(b"ACGTN", np.uint8)[rng.choice(5, L, p=[0.26, 0.24, 0.24, 0.25, 0.01])]
    seq = seq.copy()
    seq[500_000:500_400] |= 0x20
    s = seq.tobytes().decode()
    H.write_fasta(tmp_path / "r.fa", {"c": s})
    H.write_2bit(tmp_path / "r.2bit", {"c": s})
    n = 300_000
    fs = np.sort(rng.integers(0, L - 700, n)).astype(np.int32)
    fe = (fs + rng.integers(1, 600, n)).astype(np.int32)
    mq = rng.integers(0, 61, n).astype(np.uint8)
    st = rng.integers(0, 2, n).astype(np.uint8)
    fs[:3] = [0, 1, 2]
    fe[-1] = L
    engine.load_contig("motif_big", fs, fe, mq, st)
    from finaletoolkit_amd.reference import ReferenceGenome
    ws = np.array([0, 1000, 250_000, 0, 1_999_000, 700_000], np.int32)
    we = np.array([1000, 2000, 1_250_000, L, L, 700_000], np.int32)
    rws = list(zip(fs.tolist(), fe.tolist(), mq.tolist(), st.tolist()))
    h = k // 2
    for both, neg in ((True, False), (False, False), (False, True)):
        spec = (dict(fwd_offset=0, rev_offset=-k, guard=0, rev_oob_is_error=False) if kind == "end" else
                dict(fwd_offset=-h, rev_offset=-h, guard=h, rev_oob_is_error=False))
        want = np.stack([O.py_region_motifs(rws, s, int(a), int(b), k, kind, both, neg, 25) if (both, kind) != (True, "end")
                         else _end_both_no_raise(rws, s, int(a), int(b), k, 25) for a, b in zip(ws, we)])
        for path in ("r.fa", "r.2bit"):
            with ReferenceGenome(str(tmp_path / path)) as ref:
                rid = ref.device_image(engine, "c")
                got, nfrag, err = engine.motif_counts("motif_big", rid, ws, we, k, both_strands=both,
                                                      negative_strand=neg, quality_threshold=25, **spec)
            assert np.array_equal(got.astype(np.int64), want), (both, neg, path)
            assert err.sum() == 0 and nfrag[3] == int((mq >= 25).sum())
    engine.release("motif_big")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["end", "breakpoint"])
def test_gpu_many_equal_windows_take_the_block_kernels(engine, kind, tmp_path):
    """400 tiles of 5 kb (one block per window, no planner): the streamed word form of the motif pass and the general
    form (FASTA text) against the oracle, window by window; N blocks by the hundred, both strand rules."""
    k = 4
    rng = np.random.default_rng(77)
    L = 2_000_000
    seq = np.frombuffer(b"ACGTN", np.uint8)[rng.choice(5, L, p=[0.26, 0.24, 0.24, 0.255, 0.005])].copy()
    seq[1_200_000:1_230_000] = ord("N")
    s = seq.tobytes().decode()
    H.write_fasta(tmp_path / "r.fa", {"c": s})
    H.write_2bit(tmp_path / "r.2bit", {"c": s})
    n = 60_000
    fs = np.sort(rng.integers(0, L - 400, n)).astype(np.int32)
    fe = (fs + rng.integers(1, 400, n)).astype(np.int32)
    mq = rng.integers(0, 61, n).astype(np.uint8)
    st = rng.integers(0, 2, n).astype(np.uint8)
    engine.load_contig("motif_tiles", fs, fe, mq, st)
    from finaletoolkit_amd.reference import ReferenceGenome
    ws = np.arange(0, L, 5000, dtype=np.int32)
    we = (ws + 5000).astype(np.int32)
    h = k // 2
    spec = (dict(fwd_offset=0, rev_offset=-k, guard=0, rev_oob_is_error=False) if kind == "end" else
            dict(fwd_offset=-h, rev_offset=-h, guard=h, rev_oob_is_error=False))
    for both, neg in ((False, False), (False, True)):
        want = []
        for a, b in zip(ws.tolist(), we.tolist()):
            lo, hi = np.searchsorted(fs, a - 400), np.searchsorted(fs, b)
            rws = list(zip(fs[lo:hi].tolist(), fe[lo:hi].tolist(), mq[lo:hi].tolist(), st[lo:hi].tolist()))
            want.append(O.py_region_motifs(rws, s, a, b, k, kind, both, neg, 25))
        want = np.stack(want)
        for path in ("r.fa", "r.2bit"):
            with ReferenceGenome(str(tmp_path / path)) as ref:
                rid = ref.device_image(engine, "c")
                got, nfrag, err = engine.motif_counts("motif_tiles", rid, ws, we, k, both_strands=both,
                                                      negative_strand=neg, quality_threshold=25, **spec)
            assert np.array_equal(got.astype(np.int64), want), (both, neg, path)
            assert err.sum() == 0
    engine.release("motif_tiles")


def _end_both_no_raise(rws, s, a, b, k, q):
    """both-strands end motifs where no 3' k-mer leaves the contig (fe >= k holds for all rows here
    except possibly tiny fragments at the contig start: those raise in the reference, so they are
    counted by the strand-wise calls instead)."""
    fwd = O.py_region_motifs([(x, y, m, 1) for x, y, m, _ in rws], s, a, b, k, "end", False, False, q)
    rev = O.py_region_motifs([r for r in rws if r[1] - k >= 0], s, a, b, k, "end", False, True, q)
    return fwd + rev
