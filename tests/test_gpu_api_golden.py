"""
GPU: the reference-shaped Python surface (finaletoolkit_amd.frag / utils / cli)
on the committed fixtures, against the vectors the reference itself produced
(tests/golden/, see oracle/gen_golden.py).  Reads like the reference's own
tests/test_coverage.py, test_wps.py, test_frag_length.py, test_frag_io.py.
"""
import gzip
import os
import subprocess
import sys

import numpy as np
import pandas as pd
import pytest

from finaletoolkit_amd import frag
from finaletoolkit_amd.exceptions import InvalidInputError, MissingIndexError, UnsupportedFormatError
from finaletoolkit_amd.utils import frag_array, frag_generator
from tests.helpers import DATA, GOLDEN, ROOT, golden_json, golden_npz

pytestmark = pytest.mark.gpu

FIX = os.path.join(DATA, "12.3444.b37.frag.gz")
FIX_BED6 = os.path.join(DATA, "12.3444.b37.frag.bed.gz")
BAM = os.path.join(DATA, "12.3444.b37.bam")
INTERVALS = os.path.join(DATA, "intervals.bed")
SYN = os.path.join(GOLDEN, "synth.frag.gz")
WBED = os.path.join(GOLDEN, "synth_windows.bed")


@pytest.fixture(scope="module")
def G():
    return golden_json()


@pytest.fixture(scope="module")
def A():
    return golden_npz()


def _approx_stats(got, want):
    assert list(got[:4]) == list(want[:4])
    assert got[5] == want[5] and list(got[7:10]) == list(want[7:10])
    assert got[4] == pytest.approx(want[4], rel=1e-12)
    assert got[6] == pytest.approx(want[6], rel=1e-9)      # stdev: summation order differs
    assert got[10] == pytest.approx(want[10], rel=1e-12)


class TestFragIO:  # reference tests/test_frag_io.py
    def test_frag_gz(self, G):
        frags = list(frag_generator(FIX, "12", quality_threshold=0, min_length=0, max_length=9999))
        assert len(frags) == 17
        assert [list(f) for f in frags] == G["fixture"]["frag_generator_all"]

    def test_bed_gz_warns(self, G):
        with pytest.warns(UserWarning):
            frags = list(frag_generator(FIX_BED6, "12", quality_threshold=0, min_length=0, max_length=9999))
        assert [list(f) for f in frags] == G["fixture"]["frag_generator_bed6"]

    def test_bam(self):
        frags = list(frag_generator(BAM, "12", quality_threshold=0, min_length=0, max_length=9999))
        assert len(frags) == 17
        assert all(34442500 < f[1] < 34446500 for f in frags)

    def test_detailed(self, G):
        g = list(frag_generator(FIX, contig="12", start=34443119, stop=34443538))
        assert g == [("12", 34443118, 34443284, 60, True), ("12", 34443139, 34443300, 60, True),
                     ("12", 34443294, 34443491, 60, True), ("12", 34443358, 34443538, 60, False)]

    def test_frag_array(self, G):
        arr = frag_array(FIX, "12", min_length=120, max_length=180)
        assert arr.dtype == np.dtype([("start", "<i8"), ("stop", "<i8"), ("strand", "?")])
        assert [[int(r["start"]), int(r["stop"]), bool(r["strand"])] for r in arr] == G["fixture"]["frag_array_120_180"]

    def test_frag_array_and_frags_in_region_known_answers(self):  # reference tests/test_utils.py:26-60
        from finaletoolkit_amd.utils import frags_in_region
        dt = [("start", "<i8"), ("stop", "<i8"), ("strand", "?")]
        want = np.array([(34443118, 34443284, True), (34443139, 34443300, True), (34443358, 34443538, False),
                         (34443483, 34443660, True), (34444089, 34444252, True), (34444696, 34444863, True),
                         (34444954, 34445075, True), (34444968, 34445105, True), (34445136, 34445288, True),
                         (34445511, 34445672, False), (34445705, 34445852, True), (34445723, 34445893, True),
                         (34446126, 34446261, False), (34446486, 34446653, True)], dtype=dt)
        arr = frag_array(FIX, "12", min_length=120, max_length=180)
        assert np.array_equal(arr, want)
        assert np.array_equal(frags_in_region(arr, 34443119, 34445075), want[:8])

    def test_errors(self, tmp_path):
        with pytest.raises(InvalidInputError):
            list(frag_generator(FIX, None, start=5, stop=10))
        with pytest.raises(InvalidInputError):
            list(frag_generator(FIX, "12", intersect_policy="nope"))
        with pytest.raises(FileNotFoundError):
            frag.single_coverage(str(tmp_path / "missing.frag.gz"), "12")
        p = tmp_path / "noindex.frag.gz"
        p.write_bytes(open(FIX, "rb").read())
        with pytest.raises(MissingIndexError):
            frag.single_coverage(str(p), "12")
        q = tmp_path / "x.txt"
        q.write_text("12\t1\t2\n")
        with pytest.raises(UnsupportedFormatError):
            frag.single_coverage(str(q), "12")
        with pytest.raises(ValueError):
            frag.single_coverage(FIX, "chrNope", 0, 100)


class TestCoverage:  # reference tests/test_coverage.py
    def test_bam_whole_contig(self):
        chrom, start, stop, name, cov = frag.single_coverage(BAM, "12", 0, None, quality_threshold=0)
        assert (chrom, start, cov) == ("12", 0, 17)

    def test_bam_interval(self):
        assert frag.single_coverage(BAM, "12", 34443000, 34447000, quality_threshold=0).coverage == 17
        assert frag.single_coverage(BAM, "12", 34443400, 34443600, quality_threshold=0).coverage == 2

    def test_fixture_single_coverage_golden(self, G):
        for c in G["fixture"]["single_coverage"]:
            r = frag.single_coverage(FIX, "12", c["start"], c["stop"], quality_threshold=c["q"],
                                     intersect_policy=c["policy"], min_length=c["min_length"],
                                     max_length=c["max_length"])
            assert r.coverage == c["coverage"] and isinstance(r.coverage, int)

    def test_coverage_normalize(self, G):
        results = frag.coverage(FIX, INTERVALS, None, scale_factor=1., normalize=True)
        assert [list(r) for r in results] == G["fixture"]["coverage_norm"]
        assert results[0].coverage == pytest.approx(4 / 16) and results[1].coverage == pytest.approx(7 / 16)

    def test_coverage_no_normalize(self, G):
        results = frag.coverage(FIX, INTERVALS, None, normalize=False, intersect_policy="midpoint", scale_factor=1.)
        assert [list(r) for r in results] == G["fixture"]["coverage_raw"]

    def test_coverage_writers(self, G, tmp_path):
        out = str(tmp_path / "c.bed")
        frag.coverage(FIX, INTERVALS, out, normalize=True, scale_factor=1e6)
        assert open(out).read() == G["fixture"]["coverage_norm_bed_text"]
        out = str(tmp_path / "c.bedgraph")
        frag.coverage(FIX, INTERVALS, out, normalize=False)
        assert open(out).read() == G["fixture"]["coverage_bedgraph_text"]
        out = str(tmp_path / "c.bed.gz")
        frag.coverage(FIX, INTERVALS, out, normalize=True, scale_factor=1e6)
        assert gzip.open(out, "rt").read() == G["fixture"]["coverage_norm_bed_text"]
        with pytest.raises(ValueError):
            frag.coverage(FIX, INTERVALS, str(tmp_path / "c.txt"))

    def test_synth_all_variants(self, G):
        variants = {
            "default": {},
            "any_q0": dict(intersect_policy="any", quality_threshold=0),
            "len_120_180": dict(min_length=120, max_length=180),
            "q60_max150_any": dict(quality_threshold=60, max_length=150, intersect_policy="any"),
            "min300": dict(min_length=300, quality_threshold=10),
        }
        for key, kw in variants.items():
            got = frag.coverage(SYN, WBED, None, **kw)
            assert [list(r) for r in got] == G["synth"]["coverage"][key], key
        got = frag.coverage(SYN, WBED, None, normalize=True, scale_factor=1e6)
        assert [list(r) for r in got] == G["synth"]["coverage"]["normalized"]
        assert frag.single_coverage(SYN, "chrA", 0, None).coverage == G["synth"]["single_coverage_whole_chrA"]
        assert frag.single_coverage(SYN, None, 0, None).coverage == G["synth"]["single_coverage_whole_file"]

    def test_cli_coverage_smoke(self):  # reference tests/test_cli.py:155-179
        r = subprocess.run([sys.executable, "-m", "finaletoolkit_amd.cli", "coverage", FIX, INTERVALS, "--normalize",
                            "-o", "-"], capture_output=True, text=True, cwd=ROOT)
        assert r.returncode == 0, r.stderr
        assert r.stdout.splitlines() == ["12\t34443118\t34443538\t.\t0.25", "12\t34444968\t34446115\t.\t0.4375"]

    def test_config1_ten_windows(self, G):
        # BASELINE.json config 1: ten 400 bp windows tiling 12:34443000-34447000
        starts = list(range(34443000, 34447000, 400))
        got = [frag.single_coverage(FIX, "12", s, s + 400, quality_threshold=0).coverage for s in starts]
        assert sum(got) == 17 and len(got) == 10


class TestWPS:  # reference tests/test_wps.py
    def test_lwps(self):
        results = frag.wps(BAM, "12", 34444145, 34444155, 133851895, quality_threshold=0)
        assert np.all(results["contig"] == "12")
        assert np.all(results["start"] == np.arange(34444145, 34444155))
        assert np.all(results["wps"] == [-1, -1, -1, -1, -1, 1, 1, 1, 1, 1])
        assert results.dtype == np.dtype([("contig", "U16"), ("start", "i8"), ("wps", "i8")])

    def test_fixture_long(self, A):
        r = frag.wps(FIX, "12", 34443000, 34447000, 133851895, quality_threshold=0)
        assert np.array_equal(r["wps"], A["fixture_wps_34443000_34447000"])

    def test_synth_cases(self, G, A):
        for c in G["synth"]["wps_cases"]:
            r = frag.wps(SYN, c["contig"], c["start"], c["stop"], G["synth"]["contigs"][c["contig"]],
                         window_size=c["window_size"], min_length=c["min_length"], max_length=c["max_length"],
                         quality_threshold=c["quality_threshold"])
            assert np.array_equal(r["wps"], A[c["key"]]), c

    def test_degenerate_and_aliases(self):
        with pytest.warns(UserWarning):
            assert len(frag.wps(FIX, "12", 10, 10, 133851895)) == 0
        with pytest.raises(ValueError), pytest.warns(DeprecationWarning):
            frag.wps(FIX, "12", 0, 10, 133851895, fraction_low=100)

    def test_wig_writer(self, tmp_path):
        out = str(tmp_path / "w.wig")
        r = frag.wps(FIX, "12", 34444145, 34444155, 133851895, output_file=out, quality_threshold=0)
        lines = open(out).read().splitlines()
        assert lines[0] == "fixedStep\tchrom=12\tstart=34444145\tstep=1\tspan=10"
        assert [int(x) for x in lines[1:]] == r["wps"].tolist()

    def test_multi_wps_bedgraph(self, G, A, tmp_path):
        out = str(tmp_path / "m.bed.gz")
        with pytest.warns(UserWarning):  # chrZ site is skipped
            frag.multi_wps(SYN, os.path.join(GOLDEN, "synth_sites.bed"), os.path.join(GOLDEN, "synth.chrom.sizes"),
                           out, interval_size=3000)
        rows = [l.split("\t") for l in gzip.open(out, "rt").read().splitlines()]
        assert len(rows) == G["synth"]["multi_wps_rows"]
        assert np.array_equal(np.array([int(r[1]) for r in rows]), A["multi_wps_pos"])
        assert np.array_equal(np.array([int(r[3]) for r in rows]), A["multi_wps_val"])
        runs = []
        for r in rows:
            if not runs or runs[-1][0] != r[0]:
                runs.append([r[0], 0])
            runs[-1][1] += 1
        assert runs == G["synth"]["multi_wps_contig_runs"]
        assert all(int(r[2]) == int(r[1]) + 1 for r in rows[::97])


    def test_multi_wps_bigwig_all_chroms_present(self, tmp_path):
        """reference tests/test_wps.py:90-135: BED sorted alphabetically ("10" before "2") while the
        header order is ("2", "10") must not drop a chromosome; one 160 bp fragment per contig gives
        WPS +1 on [1_000_061, 1_000_100] and -1 on the two 120 bp end ranges."""
        from finaletoolkit_amd import bgzf
        from finaletoolkit_amd.bigwig import read_bigwig
        frag = str(tmp_path / "two.frag.gz")
        one = (np.array([1_000_000], np.int32), np.array([1_000_160], np.int32), np.array([60], np.uint8),
               np.array([1], np.uint8))
        bgzf.write_frag_gz(frag, [("2",) + one, ("10",) + one])
        cs = tmp_path / "cs.sizes"
        cs.write_text("2\t100000000\n10\t100000000\n")
        bed = tmp_path / "sites.bed"
        bed.write_text("10\t999500\t1000500\n2\t999500\t1000500\n")
        out = str(tmp_path / "out.bw")
        assert frag.endswith(".gz")
        frag_ret = __import__("finaletoolkit_amd").frag.multi_wps(frag, str(bed), str(cs), out, interval_size=1000,
                                                                  min_length=120, max_length=180,
                                                                  quality_threshold=0)
        assert frag_ret == out
        chroms, iv = read_bigwig(out)
        assert chroms == {"2": (0, 100000000), "10": (1, 100000000)}
        for c in ("2", "10"):
            vals = {s: v for (cc, s, e, v) in iv if cc == c}
            assert len(vals) == 1000 and min(vals) == 999_500
            assert max(vals.values()) == 1.0 and min(vals.values()) == -1.0
            assert [s for s, v in vals.items() if v == 1.0] == list(range(1_000_061, 1_000_101))
        assert [c for c, *_ in iv][0] == "2"  # header order


class TestFragLength:  # reference tests/test_frag_length.py
    def test_frag_lengths(self, G, A):
        lengths = frag.frag_length(FIX, contig="12", start=34443119, stop=34443538)
        assert lengths.dtype == np.int32 and lengths.tolist() == G["fixture"]["frag_length"]
        got = frag.frag_length(SYN, contig="chrB", start=10_000, stop=30_000, intersect_policy="any",
                               quality_threshold=0)
        assert np.array_equal(got, A["synth_frag_length_chrB_any"])
        assert np.array_equal(frag.frag_length(SYN, contig="chrA"), A["synth_frag_length_chrA_all"])

    def test_bins(self, G, tmp_path):
        bins, counts = frag.frag_length_bins(FIX, contig="12", start=34443119, stop=34443538)
        want = G["fixture"]["frag_length_bins"]
        assert np.asarray(bins).tolist() == want["bins"] and counts == want["counts"]
        bins, counts = frag.frag_length_bins(FIX, contig="12", bin_size=5, quality_threshold=0)
        want = G["fixture"]["frag_length_bins_bs5_q0"]
        assert np.asarray(bins).tolist() == want["bins"] and counts == want["counts"]
        for key, kw in [("frag_length_bins_chrA", dict(contig="chrA")),
                        ("frag_length_bins_chrB_bs7", dict(contig="chrB", start=20_000, stop=90_000, bin_size=7,
                                                           min_length=50, max_length=450)),
                        ("frag_length_bins_genome", {})]:
            bins, counts = frag.frag_length_bins(SYN, **kw)
            assert np.asarray(bins).tolist() == G["synth"][key]["bins"] and counts == G["synth"][key]["counts"], key
        out = str(tmp_path / "b.tsv")
        frag.frag_length_bins(FIX, contig="12", output_file=out, summary_stats=True, short_fraction=150)
        got, want = open(out).read().splitlines(), G["fixture"]["frag_length_bins_tsv_text"].splitlines()
        assert [l for l in got if not l.startswith("#stdev")] == [l for l in want if not l.startswith("#stdev")]
        sd = [float(l.split(": ")[1]) for l in got if l.startswith("#stdev")][0]
        assert sd == pytest.approx([float(l.split(": ")[1]) for l in want if l.startswith("#stdev")][0], rel=1e-9)
        with pytest.warns(RuntimeWarning):
            b, c = frag.frag_length_bins(FIX, contig="12", start=1, stop=2)
        assert len(b) == 0 and len(c) == 0

    def test_intervals(self, G, tmp_path):
        got = frag.frag_length_intervals(FIX, INTERVALS)
        for g, w in zip(got, G["fixture"]["frag_length_intervals"]):
            _approx_stats(g, w)
        assert got[1].median == 147.0  # the reference's odd-count median quirk (true median 152)
        got = frag.frag_length_intervals(SYN, WBED)
        for g, w in zip(got, G["synth"]["frag_length_intervals"]):
            _approx_stats(g, w)
        got = frag.frag_length_intervals(SYN, WBED, min_length=120, max_length=400, intersect_policy="any",
                                         quality_threshold=0, short_reads=167)
        for g, w in zip(got, G["synth"]["frag_length_intervals_120_400_any"]):
            _approx_stats(g, w)
        out = str(tmp_path / "iv.bed")
        frag.frag_length_intervals(FIX, INTERVALS, output_file=out, quality_threshold=0, short_reads=160)
        got_l = open(out).read().splitlines()
        want_l = G["fixture"]["frag_length_intervals_bed_text"].splitlines()
        assert got_l[0] == want_l[0] and len(got_l) == len(want_l)
        for a, b in zip(got_l[1:], want_l[1:]):
            fa, fb = a.split("\t"), b.split("\t")
            assert fa[:4] == fb[:4] and fa[7:10] == fb[7:10]
            assert [float(x) for x in fa[4:]] == pytest.approx([float(x) for x in fb[4:]], rel=1e-9)


class TestDelfi:
    def test_single_window_counts_golden(self, G):
        """Per-bin short/long/num_frags/arm/gc vs the reference's _delfi_single_window
        through the full driver (synthetic reference sequence ACGT... => gc 0.5)."""
        import finaletoolkit_amd.frag._delfi as D
        from finaletoolkit_amd.genome.gaps import ContigGaps, GenomeGaps

        class Ref:
            chroms = dict(G["synth"]["contigs"])

            def gc_count(self, contig, start, stop):
                n = stop - start
                return 2 * (n // 4) + n % 4  # "ACGT" * k + "G" * rest

            def gc_counts(self, eng, contig, starts, stops):
                return np.array([self.gc_count(contig, int(a), int(b)) for a, b in zip(starts, stops)], np.int64)

            def __enter__(self):
                return self

            def __exit__(self, *a):
                pass

        from finaletoolkit_amd.source import get_engine, open_source
        src, eng = open_source(SYN), get_engine()
        bl = D._load_blacklist_indexed(os.path.join(GOLDEN, "synth_blacklist.bed"))
        gaps = {k: ContigGaps(k, tuple(v["centromere"]), [tuple(t) for t in v["telomeres"]], v["has_short_arm"])
                for k, v in G["synth"]["gaps"].items()}
        for use_gaps in (True, False):
            for use_bl in (True, False):
                want = [r for r in G["synth"]["delfi_windows"] if r["gaps"] == use_gaps and r["blacklist"] == use_bl]
                for contig in G["synth"]["contigs"]:
                    w = [r for r in want if r["contig"] == contig]
                    rows = D._contig_windows(src, eng, Ref(), contig, np.array([r["start"] for r in w], np.int64),
                                             np.array([r["stop"] for r in w], np.int64),
                                             gaps[contig] if use_gaps else None, bl if use_bl else {}, 30)
                    for got, r in zip(rows, w):
                        assert got[:4] == (r["contig"], r["start"], r["stop"], r["arm"])
                        if r["arm"] == "NOARM":
                            assert np.isnan(got[4]) and np.isnan(got[5]) and np.isnan(got[6]) and got[7] == 0
                        else:
                            assert (got[4], got[5], got[7]) == (r["short"], r["long"], r["num_frags"])
                            if r["gc"] is None:
                                assert np.isnan(got[6])
                            else:
                                assert got[6] == pytest.approx(r["gc"], rel=1e-12)

    def test_driver_end_to_end(self, tmp_path):
        """delfi() on the synthetic file: bins -> counts -> ratio -> merge; output equals a
        recomputation from the oracle's per-window counts."""
        from oracle import oracle as O
        from tests.helpers import read_frag_gz
        size = 400_000
        bins = tmp_path / "bins.txt"
        bins.write_text("#chr\tstart\tend\n" + "".join(f"chrA\t{a}\t{a + 999}\n" for a in range(0, size, 1000)))
        cs = tmp_path / "cs.genome"
        cs.write_text(f"chrA\t{size}\n")
        fa = tmp_path / "ref.fa"
        seq = ("ACGTTGCAAT" * (size // 10))
        fa.write_text(">chrA\n" + "\n".join(seq[i:i + 60] for i in range(0, size, 60)) + "\n")
        gapbed = tmp_path / "gaps.bed"
        gapbed.write_text("chrA\t0\t10000\ttelomere\nchrA\t180000\t230000\tcentromere\nchrA\t390000\t400000\ttelomere\n"
                          "chrA\t50000\t50500\tcontig\n")
        df = frag.delfi(SYN, str(cs), str(bins), str(fa), blacklist_file=os.path.join(GOLDEN, "synth_blacklist.bed"),
                        gap_file=str(gapbed), no_gc_correct=True, remove_nocov=False, merge_bins=False)
        s, e, q, st = read_frag_gz(SYN)["chrA"]
        fr = O.Frags(s, e, q, st)
        starts = [a for a in range(0, size, 1000)
                  if not any(a < g1 and a + 999 > g0 for g0, g1 in [(0, 10000), (180000, 230000), (390000, 400000),
                                                                    (50000, 50500)])
                  and (a + 999 < 180000 or a > 230000)]  # get_arm: strictly left / right of the centromere
        bl = [l.split() for l in open(os.path.join(GOLDEN, "synth_blacklist.bed")) if l.startswith("chrA")]
        bs, be = zip(*sorted((int(x[1]), int(x[2])) for x in bl))
        sh, lg, nf = O.c_delfi_counts(fr, starts, [a + 999 for a in starts], 30, bs, be,
                                      (180000, 230000, [(0, 10000), (390000, 400000)]))
        assert df["start"].tolist() == starts
        assert df["short"].tolist() == sh.tolist() and df["long"].tolist() == lg.tolist()
        assert df["num_frags"].tolist() == nf.tolist()
        assert set(df["arm"]) == {"Ap", "Aq"}
        ratio = np.where(lg == 0, np.nan, sh / np.where(lg == 0, 1, lg))
        assert np.allclose(df["ratio"].to_numpy(), ratio, equal_nan=True, rtol=1e-12)
        gc = df["gc"].to_numpy()
        assert np.allclose(gc[nf > 0], 0.4, atol=2e-3) and np.all(np.isnan(gc[nf == 0]))
        merged = frag.delfi(SYN, str(cs), str(bins), str(fa), gap_file=str(gapbed), no_gc_correct=True,
                            remove_nocov=False, merge_bins=True, output_file=str(tmp_path / "d.tsv"))
        assert merged.shape[0] == len([1 for a in ("Ap", "Aq")]) * 0 + merged.shape[0] and merged.shape[0] >= 5
        assert open(tmp_path / "d.tsv").readline().startswith("#contig\tstart\tstop\tarm\tshort\tlong\tgc")


class TestCleavage:  # next row: reference tests/test_cleavage_profile.py
    def test_single_intervals_golden(self, G, A):
        for c in G["synth"]["cleavage_cases"]:
            r = frag.cleavage_profile(SYN, G["synth"]["contigs"][c["contig"]], c["contig"], c["start"], c["stop"],
                                      left=c["left"], right=c["right"], min_length=c["min_length"],
                                      max_length=c["max_length"], quality_threshold=c["quality_threshold"])
            assert r.dtype == np.dtype([("contig", "U16"), ("pos", "i8"), ("proportion", "f8")])
            assert np.array_equal(r["pos"], A["cleavage_pos_" + c["key"].split("_")[1]])
            assert np.array_equal(r["proportion"], A[c["key"]]), c  # float64 bit-exact

    def test_multi_bedgraph_golden(self, G, A, tmp_path):
        import hashlib
        out = str(tmp_path / "c.bed.gz")
        with pytest.warns(UserWarning):  # chrQ is not in chrom.sizes
            frag.multi_cleavage_profile(SYN, os.path.join(GOLDEN, "synth_cleavage_intervals.bed"),
                                        os.path.join(GOLDEN, "synth.chrom.sizes"), left=10, right=30, output_file=out)
        raw = gzip.open(out, "rb").read()
        rows = [l.split("\t") for l in raw.decode().splitlines()]
        assert len(rows) == G["synth"]["multi_cleavage_rows"] and rows[:3] == G["synth"]["multi_cleavage_head"]
        assert np.array_equal(np.array([int(r[1]) for r in rows]), A["multi_cleavage_pos"])
        assert np.array_equal(np.array([float(r[3]) for r in rows]), A["multi_cleavage_val"])
        assert hashlib.sha256(raw).hexdigest() == G["synth"]["multi_cleavage_text_sha"]  # text identical

    def test_vs_brute_force(self, engine):
        """The reference's own equivalence test (tests/test_cleavage_profile.py:52-86): difference-array
        result == broadcasting every fragment against every position."""
        rng = np.random.default_rng(5)
        n, size = 3000, 20_000
        s = np.sort(rng.integers(0, size - 400, n)).astype(np.int32)
        e = (s + rng.integers(1, 400, n)).astype(np.int32)
        st = (rng.random(n) < 0.5).astype(np.uint8)
        engine.load_contig("cleave_bf", s, e, np.full(n, 60, np.uint8), st)
        a, b = 4000, 16_500
        got = engine.cleavage("cleave_bf", a, b)
        pos = np.arange(a, b)[None, :]
        sel = (e > a) & (s < b)
        S, E, ST = s[sel, None].astype(np.int64), e[sel, None].astype(np.int64), st[sel, None].astype(bool)
        depth = ((S <= pos) & (pos < E)).sum(axis=0)
        ends = ((ST & (S == pos)) | (~ST & (E == pos))).sum(axis=0)
        want = np.zeros(b - a)
        want[depth != 0] = ends[depth != 0] / depth[depth != 0] * 100
        assert np.array_equal(got, want)
        engine.release("cleave_bf")


class TestDeviceGC:  # next row: DELFI per-bin GC counted on the device
    def test_2bit_and_fasta_match_host_definition(self, engine, tmp_path):
        from finaletoolkit_amd.reference import ReferenceGenome
        from tests.test_host_logic import _write_2bit
        rng = np.random.default_rng(8)
        n = 300_007
        seq = "".join(rng.choice(list("ACGT"), n))
        seq = seq[:1000] + "N" * 5003 + seq[6003:200_000] + "N" * 77 + seq[200_077:]
        tb = str(tmp_path / "g.2bit")
        _write_2bit(tb, "chrG", seq, [(1000, 5003), (200_000, 77)])
        fa = tmp_path / "g.fa"
        low = seq[:150_000] + seq[150_000:].lower()
        fa.write_text(">chrG\n" + "\n".join(low[i:i + 70] for i in range(0, n, 70)) + "\n>z\nACGT\n")
        starts = np.concatenate([np.arange(0, n - 10_000, 9_973), rng.integers(0, n - 50, 60), [0, n - 1, 999, 6002]])
        stops = np.concatenate([np.arange(0, n - 10_000, 9_973) + 10_000, starts[-64:-4] + rng.integers(1, 50, 60),
                                [n, n, 1001, 6004]])
        want = np.array([sum(ch in "GC" for ch in seq[a:b]) for a, b in zip(starts, stops)])
        for path in (tb, str(fa)):
            with ReferenceGenome(path) as ref:
                got = ref.gc_counts(engine, "chrG", starts, stops)
                assert np.array_equal(got, want), path
                assert [ref.gc_count("chrG", int(a), int(b)) for a, b in zip(starts[:40], stops[:40])] == want[:40].tolist()
