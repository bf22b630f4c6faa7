"""
TEST INFRASTRUCTURE -- runs ONLY in the build container (needs /root/reference).

BAM-mode golden vectors from the reference's OWN code.  The reference package is imported through oracle/refstub.py,
whose ``pysam.AlignmentFile`` / ``AlignedSegment`` stand-in (oracle/bamstub.py) reads BAM records with gzip + struct
and applies pysam / htslib's documented ``fetch`` and ``reference_end`` rules; everything above that -- the flag
filter, read1-only, the TLEN reconstruction (io/alignment.py:60-71,242-268), the predicates, coverage, WPS, length
statistics, DELFI windows, cleavage -- is the reference's code running.

Inputs:
    tests/data/12.3444.b37.bam      the reference's fixture (48 records, one-op and S-M-S CIGARs)
    tests/golden/edge.bam(.bai)     written here: ~5 000 pairs on three contigs whose records include soft / hard
                                    clips, I / D / N / = / X / P / B ops, read2 before read1, every filtered flag
                                    alone, TLEN 0, TLEN inconsistent with the alignment, records WITHOUT a CIGAR
                                    (TLEN > 0), CIGARs that consume no reference, fragments whose read1 lies in
                                    another window than their midpoint, fragments with a NEGATIVE start
                                    (reference_end + TLEN < 0; contig chrN), unplaced reads, aux tags, long names
    tests/golden/edge_nocigar.bam   three records; a CIGAR-less read1 with TLEN < 0 makes the reference raise
                                    TypeError (None + int, io/alignment.py:257)
Outputs:
    tests/golden/bam.json.gz, tests/golden/bam.npz, tests/golden/edge_intervals.bed, edge_sites.bed, edge.chrom.sizes

Usage:  python oracle/gen_golden_bam.py
"""
from __future__ import annotations

import gzip
import json
import os
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import refstub  # noqa: E402

refstub.install()

import finaletoolkit.frag as F  # noqa: E402  (the reference)
import finaletoolkit.frag._delfi as RD  # noqa: E402
from finaletoolkit.genome.gaps import ContigGaps  # noqa: E402
from finaletoolkit.io.alignment import AlignmentWrapper  # noqa: E402
from finaletoolkit.utils import frag_array, frag_generator  # noqa: E402

from tests import helpers as H  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "tests", "data")
FIX = os.path.join(DATA, "12.3444.b37.bam")
EDGE = os.path.join(GOLD, "edge.bam")
NOCIGAR = os.path.join(GOLD, "edge_nocigar.bam")
CONTIGS = [("chrA", 400_000), ("chrB", 150_000), ("chrN", 60_000), ("chrZ", 10_000)]
SIZES = dict(CONTIGS)

# CIGAR shapes of a read of query length rl: (text builder, reference length as a function of rl)
SHAPES = [
    (lambda rl: f"{rl}M", lambda rl: rl),
    (lambda rl: f"5S{rl - 5}M", lambda rl: rl - 5),
    (lambda rl: f"{rl - 8}M8S", lambda rl: rl - 8),
    (lambda rl: f"3S{rl - 7}M4S", lambda rl: rl - 7),
    (lambda rl: f"20M3I{rl - 23}M", lambda rl: rl - 3),
    (lambda rl: f"20M5D{rl - 20}M", lambda rl: rl + 5),
    (lambda rl: f"15M200N{rl - 15}M", lambda rl: rl + 200),
    (lambda rl: f"3H{rl}M2H", lambda rl: rl),
    (lambda rl: f"10=1X{rl - 11}=", lambda rl: rl),
    (lambda rl: f"12M2P2I{rl - 14}M", lambda rl: rl - 2),
    (lambda rl: f"4S10M1D10M2I{rl - 26}M", lambda rl: rl - 5),
    (lambda rl: f"{rl - 2}M1B2M", lambda rl: rl),  # op 9 (B) consumes nothing
]


def edge_records():
    """The edge BAM's records: [(ref_id, pos, bytes)] in file order (coordinate sorted, ties in creation order)."""
    rng = np.random.default_rng(20261003)
    recs = []  # (ref_id, pos, seq no, bytes)
    n = [0]

    def add(ref_id, pos, mapq, flag, cigar, tlen, name, **kw):
        recs.append((ref_id, pos, n[0], H.bam_record(ref_id, pos, mapq, flag, cigar, tlen, name, **kw)))
        n[0] += 1

    def shape(rl, plain_share=0.6):
        k = 0 if rng.random() < plain_share else int(rng.integers(1, len(SHAPES)))
        return SHAPES[k][0](rl), SHAPES[k][1](rl)

    def pair(ref_id, fs, fe, mq, fwd, tag, odd_tlen=0, aux=b""):
        ln = fe - fs
        rl = int(min(max(ln, 30), rng.integers(36, 76)))
        cl, ref_l = shape(rl)
        cr, ref_r = shape(rl)
        lpos, rpos = fs, fe - ref_r  # the right read ENDS at fe
        if rpos < 0:
            cr, ref_r = f"{min(rl, fe)}M", min(rl, fe)
            rpos = fe - ref_r
        tl = ln + odd_tlen
        if fwd:  # read1 = left / forward (99), read2 = right / reverse (147)
            add(ref_id, lpos, mq, 99, cl, tl, tag, mate_pos=rpos, aux=aux)
            add(ref_id, rpos, mq, 147, cr, -tl, tag, mate_pos=lpos, aux=aux)
        else:    # read1 = right / reverse (83), read2 = left / forward (163): read2 comes first in the file
            add(ref_id, rpos, mq, 83, cr, -tl, tag, mate_pos=lpos, aux=aux)
            add(ref_id, lpos, mq, 163, cl, tl, tag, mate_pos=rpos, aux=aux)

    def lengths(k):
        u = rng.random(k)
        ln = np.where(u < 0.75, rng.normal(167, 14, k), np.where(u < 0.87, rng.normal(334, 25, k),
                      np.where(u < 0.95, rng.integers(20, 120, k), rng.integers(30, 800, k))))
        return np.clip(np.rint(ln), 20, 1000).astype(np.int64)

    for ref_id, (name, size) in enumerate(CONTIGS[:3]):
        k = {"chrA": 3500, "chrB": 1300, "chrN": 220}[name]
        fs = np.sort(rng.integers(0, size - 1000, k))
        if name != "chrN":
            fs[:6] = [0, 0, 1, 3, 40, 40]  # fragments at the contig's first bases, position ties
        else:
            fs = fs + 300
        ln = lengths(k)
        mq = np.where(rng.random(k) < 0.8, 60, rng.integers(0, 60, k))
        mq[rng.random(k) < 0.01] = 255
        fwd = rng.random(k) < 0.5
        for i in range(k):
            s, e = int(fs[i]), int(min(fs[i] + ln[i], size))
            odd = int(rng.integers(-9, 10)) if rng.random() < 0.03 else 0
            aux = b"NMC\x00ASC\x28" if i % 5 == 0 else b""
            tag = f"{name}.{i}" + ("x" * 200 if i % 97 == 0 else "")
            pair(ref_id, s, e, int(mq[i]), bool(fwd[i]), tag, odd, aux)
            if i % 11 == 0:  # every rejected kind alone, on a read1-shaped record that would otherwise count
                rl = min(e - s, 50)
                for j, (flag, tl) in enumerate([(99 | 0x400, e - s), (99 | 0x100, e - s), (99 | 0x200, e - s),
                                                (99 | 0x800, e - s), (97, e - s), (99 | 0x8, e - s), (0x40 | 0x20, e - s),
                                                (99, 0), (0x4 | 0x1 | 0x2 | 0x40 | 0x20, e - s), (99 | 0x80, e - s),
                                                (83 | 0x400, -(e - s)), (81, -(e - s))]):
                    add(ref_id, s, int(mq[i]), flag, f"{rl}M", tl, f"{name}.j{i}.{j}", mate_pos=s)
            if i % 53 == 0:  # neither read1 nor read2 flagged: NOT read2, so the reference keeps it
                add(ref_id, s, int(mq[i]), 3 | 0x20, f"{min(e - s, 40)}M", e - s, f"{name}.n{i}", mate_pos=s)

    # --- hand-made records ---------------------------------------------------------------------------------------
    A, B, N = 0, 1, 2
    # no CIGAR, TLEN > 0: the reference needs no reference_end here and yields [pos, pos + tlen)
    add(A, 120_000, 60, 99, "*", 170, "nocigar.fwd", l_seq=0, mate_pos=120_120)
    add(A, 120_120, 60, 147, "50M", -170, "nocigar.fwd", mate_pos=120_000)
    add(B, 9_990, 47, 99, "*", 160, "nocigar.edge", l_seq=40, mate_pos=10_100)  # read1 "spans" one base: [9990, 9991)
    # CIGARs that consume no reference: reference_end = pos + 1 (htslib counts an empty alignment as one base)
    add(A, 130_000, 60, 83, "30S", -150, "noref.rev", mate_pos=129_851)          # -> [129851, 130001)
    add(A, 129_851, 60, 163, "40M", 150, "noref.rev", mate_pos=130_000)
    add(A, 131_000, 55, 99, "10I20S", 140, "noref.fwd", mate_pos=131_090)        # -> [131000, 131140)
    add(B, 19_999, 60, 83, "25S", -180, "noref.rev.edge", mate_pos=19_820)       # read1 = [19999, 20000): window edge
    # long fragments whose read1 lies in another 10 kb window than their midpoint
    add(A, 49_800, 60, 99, "60M", 900, "far.fwd", mate_pos=50_640)               # read1 [49800,49860), midpoint 50250
    add(A, 50_640, 60, 147, "60M", -900, "far.fwd", mate_pos=49_800)
    add(A, 60_300, 60, 83, "50M", -800, "far.rev", mate_pos=59_550)              # read1 [60300,60350), frag [59550,60350)
    add(A, 59_550, 60, 163, "50M", 800, "far.rev", mate_pos=60_300)
    add(B, 29_990, 33, 99, "20M", 400, "far.fwd.b", mate_pos=30_370)             # read1 ends exactly at the bound
    add(B, 30_000, 33, 83, "20M", -400, "far.rev.b", mate_pos=29_620)            # read1 starts exactly at the bound
    # a read1 with a 3 kb N skip: the alignment overlaps windows its fragment (by TLEN) does not reach
    add(A, 150_000, 60, 99, "20M3000N30M", 180, "skip.fwd", mate_pos=150_130)
    add(A, 153_100, 60, 83, "20M3000N30M", -200, "skip.rev", mate_pos=155_950)   # ref_end 156150 -> [155950, 156150)
    # negative starts (chrN): reference_end + TLEN < 0
    add(N, 10, 60, 83, "40M", -120, "neg.a", mate_pos=0)                          # -> [-70, 50)
    add(N, 0, 60, 83, "5S30M", -31, "neg.b", mate_pos=0)                          # -> [-1, 30)
    add(N, 100, 42, 83, "60M", -400, "neg.c", mate_pos=0)                         # -> [-240, 160)
    add(N, 5, 60, 83, "20M", -25, "neg.zero", mate_pos=0)                         # -> [0, 25): not negative
    add(N, 200, 60, 83, "50M", -150_000, "neg.huge", mate_pos=0)                  # -> [-149750, 250)
    # unplaced pair at the end of the file
    recs.sort(key=lambda r: (r[0], r[1], r[2]))
    out = [(r[0], r[1], r[3]) for r in recs]
    out.append((-1, -1, H.bam_record(-1, -1, 0, 77, "*", 0, "unplaced", l_seq=30, mate_ref=-1, mate_pos=-1)))
    out.append((-1, -1, H.bam_record(-1, -1, 0, 141, "*", 0, "unplaced", l_seq=30, mate_ref=-1, mate_pos=-1)))
    return out


def nocigar_records():
    return [(0, 1_000, H.bam_record(0, 1_000, 60, 99, "50M", 170, "ok", mate_pos=1_120)),
            (0, 1_120, H.bam_record(0, 1_120, 60, 147, "50M", -170, "ok", mate_pos=1_000)),
            (0, 2_000, H.bam_record(0, 2_000, 60, 83, "*", -160, "nocigar.rev", l_seq=0, mate_pos=1_840))]


def rows(frags):
    return [[f[0], int(f[1]), int(f[2]), int(f[3]), bool(f[4])] for f in frags]


def tup(x):
    return [None if v is None else (v.item() if hasattr(v, "item") else v) for v in x]


def attempt(fn):
    """The value, or the exception type the reference raises."""
    try:
        return dict(ok=True, value=fn())
    except Exception as e:  # noqa: BLE001 - the error type is the golden
        return dict(ok=False, error=type(e).__name__, message=str(e)[:160])


def bam_cases(path, contigs, wins, wps_cases, A, tag, delfi=None, whole_file=True):
    """Everything recorded for one BAM.  ``contigs``: {name: size}; ``wins``: [(contig, start, stop)]."""
    J = {}
    with AlignmentWrapper(path, quality_threshold=0) as aw:
        J["chroms"] = dict(aw.chroms)
        J["is_sam"] = bool(aw.is_sam)
    fetch = []
    for q in (0, 30):
        with AlignmentWrapper(path, quality_threshold=q) as aw:
            regions = ([(None, None, None)] if whole_file and q == 0 else []) + [(c, None, None) for c in contigs] + list(wins[::3])
            for c, a, b in regions:
                fetch.append(dict(quality_threshold=q, contig=c, start=a, stop=b, fragments=rows(aw.fetch(c, a, b))))
    J["fetch"] = fetch
    gen = []
    for c, a, b in wins[::2]:
        for kw in (dict(), dict(quality_threshold=0, intersect_policy="any"),
                   dict(min_length=100, max_length=220, quality_threshold=20)):
            gen.append(dict(contig=c, start=a, stop=b, kw=kw, fragments=rows(frag_generator(path, c, start=a, stop=b, **kw))))
    for c in contigs:
        gen.append(dict(contig=c, start=None, stop=None, kw=dict(quality_threshold=0),
                        fragments=rows(frag_generator(path, c, quality_threshold=0))))
    J["frag_generator"] = gen
    cov = []
    for c, a, b in wins:
        for kw in (dict(), dict(quality_threshold=0), dict(quality_threshold=0, intersect_policy="any"),
                   dict(intersect_policy="any", min_length=120, max_length=180),
                   dict(quality_threshold=60, max_length=150), dict(min_length=300, quality_threshold=10)):
            cov.append(dict(contig=c, start=a, stop=b, kw=kw, coverage=int(F.single_coverage(path, c, a, b, **kw).coverage)))
    for c in contigs:
        for kw in (dict(), dict(quality_threshold=0, intersect_policy="any")):
            cov.append(dict(contig=c, start=0, stop=None, kw=kw, coverage=int(F.single_coverage(path, c, 0, None, **kw).coverage)))
    cov.append(dict(contig=None, start=0, stop=None, kw=dict(quality_threshold=0),
                    coverage=int(F.single_coverage(path, None, 0, None, quality_threshold=0).coverage)))
    J["single_coverage"] = cov
    fa = []
    for c, a, b in wins[::4]:
        for kw in (dict(), dict(min_length=120, max_length=180, quality_threshold=0, intersect_policy="any")):
            r = frag_array(path, c, start=a, stop=b, **kw)
            fa.append(dict(contig=c, start=a, stop=b, kw=kw, rows=[[int(x["start"]), int(x["stop"]), bool(x["strand"])] for x in r]))
    J["frag_array"] = fa
    wp = []
    for k, (c, a, b, W, mn, mx, q) in enumerate(wps_cases):
        r = F.wps(path, c, a, b, contigs[c], window_size=W, min_length=mn, max_length=mx, quality_threshold=q)
        A[f"{tag}_wps_{k}"] = r["wps"].astype(np.int64)
        wp.append(dict(key=f"{tag}_wps_{k}", contig=c, start=a, stop=b, window_size=W, min_length=mn, max_length=mx,
                       quality_threshold=q))
    J["wps"] = wp
    fl = []
    for k, (c, a, b) in enumerate(wins[::5]):
        for j, kw in enumerate((dict(), dict(intersect_policy="any", quality_threshold=0))):
            A[f"{tag}_fraglen_{k}_{j}"] = np.asarray(F.frag_length(path, contig=c, start=a, stop=b, **kw))
            fl.append(dict(key=f"{tag}_fraglen_{k}_{j}", contig=c, start=a, stop=b, kw=kw))
    J["frag_length"] = fl
    bins = []
    for kw in [dict(contig=c) for c in contigs] + [dict(), dict(contig=next(iter(contigs)), bin_size=7, min_length=50,
                                                                  max_length=450, quality_threshold=0)]:
        bb, cc = F.frag_length_bins(path, **kw)
        bins.append(dict(kw=kw, bins=np.asarray(bb).tolist(), counts=list(map(int, cc))))
    J["frag_length_bins"] = bins
    cl = []
    for k, (c, a, b) in enumerate(wins[1::6]):
        r = F.cleavage_profile(path, contigs[c], c, a, min(a + 3000, b), left=5, right=10, quality_threshold=20)
        A[f"{tag}_cleavage_{k}"] = r["proportion"].astype(np.float64)
        cl.append(dict(key=f"{tag}_cleavage_{k}", contig=c, start=a, stop=min(a + 3000, b), left=5, right=10,
                       quality_threshold=20))
    J["cleavage"] = cl
    if delfi is not None:
        J["delfi_windows"] = delfi(path)
    return J


def main():
    warnings.simplefilter("ignore")
    H.write_bam(EDGE, CONTIGS, edge_records())
    H.write_bam(NOCIGAR, [("chrE", 20_000)], nocigar_records())
    J, A = {}, {}

    # ------------------------------------------------------------------ the reference's fixture
    fwins = [("12", a, a + 400) for a in range(34_443_000, 34_447_000, 400)] + [("12", 34_443_400, 34_443_600),
                                                                                 ("12", 34_444_000, 34_446_000)]
    fwps = [("12", 34_444_145, 34_444_155, 120, 120, 180, 0), ("12", 34_443_000, 34_447_000, 120, 120, 180, 0),
            ("12", 34_443_000, 34_447_000, 120, 120, 180, 30), ("12", 34_443_100, 34_446_700, 61, 30, 400, 0)]
    J["fixture"] = bam_cases(FIX, {"12": 133_851_895}, fwins, fwps, A, "fixture")
    J["fixture"]["coverage_raw"] = [list(r) for r in F.coverage(FIX, os.path.join(DATA, "intervals.bed"), None, normalize=False)]
    J["fixture"]["coverage_norm"] = [list(r) for r in F.coverage(FIX, os.path.join(DATA, "intervals.bed"), None, normalize=True)]
    J["fixture"]["frag_length_intervals"] = [tup(r) for r in F.frag_length_intervals(FIX, os.path.join(DATA, "intervals.bed"))]

    # ------------------------------------------------------------------ the edge BAM: chrA / chrB (/ empty chrZ)
    main_contigs = {"chrA": 400_000, "chrB": 150_000, "chrZ": 10_000}
    wins = []
    for c in ("chrA", "chrB"):
        wins += [(c, a, min(a + 10_000, SIZES[c])) for a in range(0, SIZES[c], 10_000)]
    wins += [("chrA", 49_860, 50_000), ("chrA", 120_000, 120_001), ("chrA", 129_990, 130_001), ("chrA", 152_000, 153_000),
             ("chrB", 9_991, 10_050), ("chrB", 19_999, 20_000), ("chrB", 20_000, 20_001), ("chrZ", 0, 10_000),
             ("chrA", 0, 1), ("chrA", 399_000, 400_000), ("chrA", 0, 400_000), ("chrB", 75_000, 75_000)]
    wps_cases = []
    for (W, mn, mx, q) in [(120, 120, 180, 30), (120, 30, 400, 0), (40, 30, 90, 30), (121, 100, 200, 30), (7, 0, 1000, 0)]:
        for (c, a, b) in [("chrA", 0, 1500), ("chrA", 398_700, 400_000), ("chrA", 119_000, 121_500),
                          ("chrA", 129_000, 131_500), ("chrA", 149_500, 157_000), ("chrB", 9_000, 11_000),
                          ("chrB", 149_000, 150_000), ("chrZ", 100, 600)]:
            wps_cases.append((c, a, b, W, mn, mx, q))

    bl = {}
    for i, c in enumerate(("chrA", "chrB")):
        r2 = np.random.default_rng(777 + i)
        s0 = np.sort(r2.integers(0, SIZES[c] - 5000, 50))
        e0 = s0 + r2.integers(150, 4000, 50)
        bl[c] = (s0.astype(np.int64), e0.astype(np.int64))
    gaps = {"chrA": ContigGaps("chrA", (180_000, 230_000), [(0, 10_000), (390_000, 400_000)]),
            "chrB": ContigGaps("chrB", (60_000, 80_000), [(0, 140_000)], has_short_arm=True)}

    class _Ref:
        chroms = dict(main_contigs)

        def sequence(self, contig, start, stop):
            return "ACGT" * ((stop - start) // 4) + "G" * ((stop - start) % 4)

    def delfi_rows(path):
        out = []
        for use_gaps in (True, False):
            for use_bl in (True, False):
                RD._WORKER_ALIGNMENT = AlignmentWrapper(path, quality_threshold=30)
                RD._WORKER_REF = _Ref()
                RD._WORKER_BLACKLIST = bl if use_bl else {}
                RD._WORKER_CONTIG_GAPS = gaps if use_gaps else None
                for c in ("chrA", "chrB"):
                    for a in range(0, SIZES[c], 10_000):
                        r = RD._delfi_single_window(c, a, a + 9_999)
                        out.append(dict(gaps=use_gaps, blacklist=use_bl, contig=r[0], start=int(r[1]), stop=int(r[2]),
                                        arm=r[3], short=None if r[4] != r[4] else int(r[4]),
                                        long=None if r[5] != r[5] else int(r[5]),
                                        gc=None if r[6] != r[6] else float(r[6]), num_frags=int(r[7])))
        return out

    e = bam_cases(EDGE, main_contigs, wins, wps_cases, A, "edge", delfi=delfi_rows)
    e["blacklist"] = {c: [v[0].tolist(), v[1].tolist()] for c, v in bl.items()}
    e["gaps"] = {k: dict(centromere=list(v.centromere), telomeres=[list(t) for t in v.telomeres],
                         has_short_arm=v.has_short_arm) for k, v in gaps.items()}
    ivals = os.path.join(GOLD, "edge_intervals.bed")
    with open(ivals, "w") as fh:
        fh.write("# windows over the edge BAM\n")
        for k, (c, a, b) in enumerate(wins):
            if b > a:
                fh.write(f"{c}\t{a}\t{b}\tw{k}\n")
    for key, kw in {"default": {}, "any_q0": dict(intersect_policy="any", quality_threshold=0),
                    "len_120_180": dict(min_length=120, max_length=180)}.items():
        e[f"coverage_{key}"] = [list(r) for r in F.coverage(EDGE, ivals, None, **kw)]
    # normalisation divides by the WHOLE file's total (contig=None): chrN's negative starts are part of it
    e["coverage_normalized"] = [list(r) for r in F.coverage(EDGE, ivals, None, normalize=True, scale_factor=1e6)]
    e["frag_length_intervals"] = [tup(r) for r in F.frag_length_intervals(EDGE, ivals)]
    e["frag_length_intervals_any_q0"] = [tup(r) for r in F.frag_length_intervals(
        EDGE, ivals, min_length=50, max_length=600, intersect_policy="any", quality_threshold=0, short_reads=167)]
    # multi_wps over a few sites -> bedGraph.gz rows
    sites = os.path.join(GOLD, "edge_sites.bed")
    with open(sites, "w") as fh:
        fh.write("chrB\t9900\t10100\nchrA\t120000\t120400\nchrA\t130000\t130100\nchrA\t399900\t400000\nchrA\t50\t60\n")
    sizes = os.path.join(GOLD, "edge.chrom.sizes")
    with open(sizes, "w") as fh:
        for c, n in CONTIGS:
            fh.write(f"{c}\t{n}\n")
    tmp = tempfile.mkdtemp(prefix="bam_gold_")
    mw = os.path.join(tmp, "mwps.bed.gz")
    F.multi_wps(EDGE, sites, sizes, mw, interval_size=2000)
    mrows = [ln.split("\t") for ln in gzip.open(mw, "rt").read().splitlines()]
    A["edge_multi_wps_pos"] = np.array([int(r[1]) for r in mrows], np.int64)
    A["edge_multi_wps_val"] = np.array([int(r[3]) for r in mrows], np.int64)
    e["multi_wps_contigs"] = sorted({r[0] for r in mrows})
    J["edge"] = e

    # ------------------------------------------------------------------ chrN: fragments with negative starts
    nwins = [("chrN", a, a + 10_000) for a in range(0, 60_000, 10_000)] + [("chrN", 0, 100), ("chrN", 0, 1), ("chrN", 25, 60_000)]
    nwps = [("chrN", 0, 800, 120, 120, 180, 0), ("chrN", 0, 800, 120, 20, 500, 0), ("chrN", 0, 400, 40, 10, 90, 30),
            ("chrN", 0, 300, 7, 0, 1000, 0), ("chrN", 5_000, 6_000, 120, 120, 180, 30)]
    J["negative_start"] = bam_cases(EDGE, {"chrN": 60_000}, nwins, nwps, A, "neg", whole_file=False)

    # ------------------------------------------------------------------ what the reference raises
    err = {}
    err["nocigar_negative_tlen_fetch"] = attempt(lambda: rows(AlignmentWrapper(NOCIGAR, quality_threshold=0).fetch("chrE")))
    err["nocigar_negative_tlen_coverage"] = attempt(lambda: int(F.single_coverage(NOCIGAR, "chrE", 0, None).coverage))
    err["nocigar_negative_tlen_region_without_it"] = attempt(
        lambda: rows(AlignmentWrapper(NOCIGAR, quality_threshold=0).fetch("chrE", 900, 1_500)))
    err["nocigar_negative_tlen_wps"] = attempt(lambda: F.wps(NOCIGAR, "chrE", 1_900, 2_100, 20_000)["wps"].tolist())
    err["unknown_contig"] = attempt(lambda: int(F.single_coverage(EDGE, "chrQ", 0, 100).coverage))
    err["negative_region_start"] = attempt(lambda: int(F.single_coverage(EDGE, "chrA", -5, 100).coverage))
    err["start_beyond_stop"] = attempt(lambda: int(F.single_coverage(EDGE, "chrA", 500, 100).coverage))
    err["bounds_without_contig"] = attempt(lambda: int(F.single_coverage(EDGE, None, 5, 100).coverage))
    err["region_beyond_contig"] = attempt(lambda: int(F.single_coverage(EDGE, "chrZ", 50_000, 60_000).coverage))
    J["errors"] = err
    for k, v in err.items():
        print("error case", k, v if not v["ok"] else ("ok", v["value"] if not isinstance(v["value"], list) else len(v["value"])))

    with open(os.path.join(GOLD, "bam.json.gz"), "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, filename="") as fh:
        fh.write(json.dumps(J, sort_keys=True, separators=(",", ":")).encode())
    np.savez_compressed(os.path.join(GOLD, "bam.npz"), **A)
    n_frag = {k: sum(len(c["fragments"]) for c in v["fetch"]) for k, v in J.items() if "fetch" in v}
    print("BAM goldens written to", GOLD, n_frag, "edge.bam", os.path.getsize(EDGE), "bytes")


if __name__ == "__main__":
    main()
