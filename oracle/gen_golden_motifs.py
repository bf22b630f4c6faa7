"""
TEST INFRASTRUCTURE -- runs ONLY in the build container (needs /root/reference).

Golden vectors for the end-motif / breakpoint-motif features: imports the reference's
``frag/_end_motifs.py`` and ``frag/_breakpoint_motifs.py`` through oracle/refstub.py (tabix and
FASTA stand-ins) and records their outputs on tests/golden/synth.frag.gz against a seeded
synthetic reference genome written here:

    tests/golden/synth_ref.fa.gz    chrA (400 kb) + chrB (150 kb): random ACGT, N runs, soft-masked
                                    (lower-case) runs, 60 bases per line  (gzip of the FASTA text)
    tests/golden/motif_intervals.bed  intervals (some without a name column)
    tests/golden/motifs.npz/.json   outputs

A second reference, the same text with chrB cut to 148 000 bases, is derived (not committed: the
tests derive it the same way) to exercise k-mers that fall off the contig.

Usage:  python oracle/gen_golden_motifs.py
"""
from __future__ import annotations

import gzip
import hashlib
import json
import os
import sys
import tempfile
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)

import refstub  # noqa: E402

refstub.install()

import finaletoolkit.frag._motif_common as MC  # noqa: E402  (the reference)
import finaletoolkit.frag._end_motifs as EM  # noqa: E402
import finaletoolkit.frag._breakpoint_motifs as BM  # noqa: E402


class _InlinePool:
    def __init__(self, n):
        pass

    def imap(self, fn, it, chunksize=1):
        return map(fn, it)

    def close(self):
        pass


MC.Pool = _InlinePool  # same results, in order, without forking

GOLD = os.path.join(ROOT, "tests", "golden")
FRAGS = os.path.join(GOLD, "synth.frag.gz")
CONTIGS = {"chrA": 400_000, "chrB": 150_000}
SHORT_B = 148_000


def make_reference():
    rng = np.random.default_rng(4242)
    seqs = {}
    for name, n in CONTIGS.items():
        s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
        for _ in range(6):  # soft-masked runs
            a = int(rng.integers(0, n - 3000))
            s[a:a + int(rng.integers(200, 3000))] |= 0x20
        for _ in range(5):  # N runs (one at the very start of chrA)
            a = int(rng.integers(0, n - 2000))
            s[a:a + int(rng.integers(1, 1500))] = ord("N")
        if name == "chrA":
            s[:5000] = ord("N")
            s[123_456] = ord("n")
        seqs[name] = s.tobytes().decode()
    return seqs


def write_fasta(path, seqs, width=60):
    with open(path, "w") as fh:
        for name, s in seqs.items():
            fh.write(f">{name}\n")
            for i in range(0, len(s), width):
                fh.write(s[i:i + width] + "\n")


def main():
    seqs = make_reference()
    tmp = tempfile.mkdtemp()
    fa = os.path.join(tmp, "synth_ref.fa")
    write_fasta(fa, seqs)
    open(fa + ".fai", "w").close()  # the stand-in needs no index; keeps the reference from calling faidx
    with open(fa, "rb") as fh, gzip.GzipFile(os.path.join(GOLD, "synth_ref.fa.gz"), "wb", mtime=0) as out:
        out.write(fh.read())
    fa_short = os.path.join(tmp, "synth_ref_short.fa")
    write_fasta(fa_short, {"chrA": seqs["chrA"], "chrB": seqs["chrB"][:SHORT_B]})
    open(fa_short + ".fai", "w").close()

    A, J = {}, {}
    warnings.simplefilter("ignore")

    def vec(d):
        return np.array(list(d.values()), np.float64)

    # genome-wide
    for key, fn, kw in [
        ("end_k4_both", EM.end_motifs, dict(k=4)),
        ("end_k3_fwd", EM.end_motifs, dict(k=3, both_strands=False, quality_threshold=0)),
        ("end_k5_neg", EM.end_motifs, dict(k=5, both_strands=False, negative_strand=True)),
        ("bp_k6_both", BM.breakpoint_motifs, dict(k=6)),
        ("bp_k4_fwd", BM.breakpoint_motifs, dict(k=4, both_strands=False, quality_threshold=10)),
        ("bp_k2_neg", BM.breakpoint_motifs, dict(k=2, both_strands=False, negative_strand=True)),
    ]:
        r = fn(FRAGS, fa, **kw)
        A[key] = np.array(r.frequencies(), np.float64)
        J[key] = dict(kw=kw, mds=r.motif_diversity_score())
    out = os.path.join(tmp, "freqs.tsv")
    EM.end_motifs(FRAGS, fa, k=4, output_file=out)
    J["end_k4_both"]["tsv_sha256"] = hashlib.sha256(open(out, "rb").read()).hexdigest()

    # per interval (the window BED of the coverage goldens + hand-picked regions)
    bed = os.path.join(GOLD, "motif_intervals.bed")   # the coverage windows without comment / track lines
    with open(os.path.join(GOLD, "synth_windows.bed")) as src, open(bed, "w") as dst:
        for line in src:
            f = line.split()
            if len(f) >= 3 and f[1].isdigit():
                dst.write("\t".join(f[:3] + ([f[3]] if len(f) > 3 and f[0] != "chrB" else [])) + "\n")
    regions = [("chrA", 0, 6000, "nrun"), ("chrA", 120_000, 130_000, "."), ("chrB", 140_000, 150_000, "tail"),
               ("chrA", 399_000, 400_000, "."), ("chrQ", 0, 100, "missing"), ("chrB", 500, 500, "empty")]
    for key, fn, kw in [
        ("iv_end_k4_both", EM.interval_end_motifs, dict(k=4)),
        ("iv_end_k2_neg_q0", EM.interval_end_motifs, dict(k=2, both_strands=False, negative_strand=True,
                                                           quality_threshold=0)),
        ("iv_bp_k4_both", BM.interval_breakpoint_motifs, dict(k=4)),
        ("iv_bp_k6_fwd", BM.interval_breakpoint_motifs, dict(k=6, both_strands=False)),
    ]:
        r = fn(FRAGS, fa, bed, **kw)
        A[key] = np.array([list(f.values()) for _, f in r.intervals], np.int64)
        J[key] = dict(kw=kw, n=len(r.intervals))
        r2 = fn(FRAGS, fa, regions, **kw)
        A[key + "_regions"] = np.array([list(f.values()) for _, f in r2.intervals], np.int64)
        if key == "iv_end_k4_both":
            out = os.path.join(tmp, "iv.tsv")
            r.to_tsv(out)
            J[key]["tsv_sha256"] = hashlib.sha256(open(out, "rb").read()).hexdigest()
            r.to_tsv(out, calc_freq=False, sep=",")
            J[key]["csv_counts_sha256"] = hashlib.sha256(open(out, "rb").read()).hexdigest()
            mds = r.motif_diversity_score()
            A["iv_end_k4_both_mds"] = np.array([m for _, m in mds], np.float64)
            A["iv_end_k4_both_mds_mm"] = np.array([m for _, m in r.motif_diversity_score(True)], np.float64)
    J["regions"] = regions
    r = BM.interval_breakpoint_motifs(FRAGS, fa, regions, k=5)
    A["iv_bp_k5_odd_regions"] = np.array([list(f.values()) for _, f in r.intervals], np.int64)

    # k-mers falling off a shorter contig (chrB cut to SHORT_B)
    tail = ("chrB", 140_000, 150_000)
    J["short_b"] = SHORT_B
    try:
        EM.region_end_motifs(FRAGS, *tail, fa_short, k=4)
        J["short_end_both"] = "no error"
    except RuntimeError:
        J["short_end_both"] = "RuntimeError"
    A["short_end_k4_fwd"] = vec(EM.region_end_motifs(FRAGS, *tail, fa_short, k=4, both_strands=False)).astype(np.int64)
    A["short_end_k4_neg"] = vec(EM.region_end_motifs(FRAGS, *tail, fa_short, k=4, both_strands=False,
                                                     negative_strand=True)).astype(np.int64)
    A["short_bp_k4_both"] = vec(BM.region_breakpoint_motifs(FRAGS, *tail, fa_short, k=4)).astype(np.int64)
    A["short_bp_k4_neg"] = vec(BM.region_breakpoint_motifs(FRAGS, *tail, fa_short, k=4, both_strands=False,
                                                           negative_strand=True)).astype(np.int64)
    with open(os.path.join(GOLD, "motifs.json"), "w") as fh:
        json.dump(J, fh, indent=0, sort_keys=True)
    np.savez_compressed(os.path.join(GOLD, "motifs.npz"), **A)
    print("motif goldens written;", {k: int(v.sum()) if v.dtype == np.int64 else round(float(v.sum()), 6)
                                     for k, v in A.items() if v.ndim <= 2 and "mds" not in k})


if __name__ == "__main__":
    main()
