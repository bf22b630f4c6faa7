"""
TEST INFRASTRUCTURE -- runs ONLY in the build container (needs /root/reference).

Pins the DELFI DRIVER to the reference: imports the reference package through oracle/refstub.py and runs its own
``finaletoolkit.frag.delfi(...)`` (``frag/_delfi.py:129-401``: gap-overlap bin filter, chrom.sizes order, NOARM
drop, ratio, the positional 8779 / 13664 drop + ``reset_index``, ``delfi_merge_bins``, ``_write_delfi``) with
``workers=1`` on ``tests/golden/synth.frag.gz`` and a synthetic FASTA, over a 20 bp bin tiling fine enough that more
than 13 665 bins survive (so the positional drop bites), with gaps on/off x blacklist on/off, merged and unmerged,
and through every writer.  Outputs (tests/golden/):

    delfi_driver.json          the cases: arguments, frame shapes, the reference's error types
    delfi_driver_<case>.csv.gz the returned frame as ``to_csv(index=False)`` text (gzip, mtime 0)
    delfi_driver_out.*         the files / stdout ``_write_delfi`` produced for one case
    synth_gaps.bed, synth_gaps_shortarm.bed, synth_bins_mixed.bed, synth_dup.chrom.sizes   (else: the existing synth.chrom.sizes)

Neither the FASTA nor the 27 500-row 20 bp bins file is committed: ``tests/helpers.synth_reference`` /
``tests/helpers.write_bins20`` regenerate them (the FASTA checked by sha256 here and in the tests).  Usage:  python oracle/gen_golden_delfi.py
"""
from __future__ import annotations

import contextlib
import gzip
import hashlib
import io
import json
import os
import sys
import tempfile
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import refstub  # noqa: E402

refstub.install()

import finaletoolkit.frag as F  # noqa: E402  (the reference)

from tests import helpers as H  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CONTIGS = {"chrA": 400_000, "chrB": 150_000}


def write_inputs():
    # mixed sizes, a bin beyond the contig end (GC skipped with a warning), a contig chrom.sizes lacks
    with open(os.path.join(GOLD, "synth_bins_mixed.bed"), "w") as fh:
        for c in ("chrA", "chrB"):
            for a in range(0, CONTIGS[c] + 20_000, 10_000):
                fh.write(f"{c}\t{a}\t{a + 9_999}\textra\n")
        fh.write("chrZ\t0\t9999\n")
    with open(os.path.join(GOLD, "synth_dup.chrom.sizes"), "w") as fh:  # a contig listed twice: its bins once per listing
        fh.write("chrB\t150000\nchrA\t400000\nchrB\t150000\n")
    with open(os.path.join(GOLD, "synth_gaps.bed"), "w") as fh:
        fh.write("chrA\t0\t10000\ttelomere\nchrA\t180000\t230000\tcentromere\nchrA\t390000\t400000\ttelomere\n"
                 "chrB\t0\t5000\ttelomere\nchrB\t60000\t80000\tcentromere\nchrB\t145000\t150000\ttelomere\n")
    with open(os.path.join(GOLD, "synth_gaps_shortarm.bed"), "w") as fh:
        fh.write("chrA\t0\t10000\ttelomere\nchrA\t180000\t230000\tcentromere\nchrA\t390000\t400000\ttelomere\n"
                 "chrB\t0\t60000\tshort_arm\nchrB\t60000\t80000\tcentromere\nchrB\t145000\t150000\ttelomere\n")


def frame_text(df):
    return df.to_csv(index=False)


def gz_write(path, text):
    with open(path, "wb") as raw, gzip.GzipFile(fileobj=raw, mode="wb", mtime=0, filename="") as fh:
        fh.write(text.encode())


def main():
    warnings.simplefilter("ignore")
    write_inputs()
    tmp = tempfile.mkdtemp(prefix="delfi_gold_")
    fasta = os.path.join(tmp, "synth_ref.fa")
    sha = H.synth_reference(fasta, CONTIGS)
    frag = os.path.join(GOLD, "synth.frag.gz")
    H.write_bins20(os.path.join(tmp, "synth_bins20.bed"), CONTIGS)
    g = lambda name: os.path.join(tmp if name == "synth_bins20.bed" else GOLD, name)  # noqa: E731
    cases = {
        "bins20_gaps_bl": dict(bins="synth_bins20.bed", gaps="synth_gaps.bed", bl=True, merge=False, nocov=True),
        "bins20_plain": dict(bins="synth_bins20.bed", gaps=None, bl=False, merge=False, nocov=True),
        "bins20_keep_nocov": dict(bins="synth_bins20.bed", gaps="synth_gaps.bed", bl=True, merge=False, nocov=False),
        "bins20_merged": dict(bins="synth_bins20.bed", gaps="synth_gaps.bed", bl=True, merge=True, nocov=True),
        "bins20_merged_keep": dict(bins="synth_bins20.bed", gaps="synth_gaps.bed", bl=False, merge=True, nocov=False),
        "bins20_shortarm": dict(bins="synth_bins20.bed", gaps="synth_gaps_shortarm.bed", bl=True, merge=True, nocov=True),
        "mixed_gaps": dict(bins="synth_bins_mixed.bed", gaps="synth_gaps.bed", bl=True, merge=False, nocov=True),
        "mixed_plain_q0": dict(bins="synth_bins_mixed.bed", gaps=None, bl=False, merge=False, nocov=False, q=0),
        "mixed_dup_listing": dict(bins="synth_bins_mixed.bed", gaps="synth_gaps.bed", bl=True, merge=False, nocov=True,
                                  sizes="synth_dup.chrom.sizes"),
    }
    J = {"fasta_sha256": sha, "cases": {}}
    for name, c in cases.items():
        df = F.delfi(frag, g(c.get("sizes", "synth.chrom.sizes")), g(c["bins"]), fasta,
                     blacklist_file=g("synth_blacklist.bed") if c["bl"] else None,
                     gap_file=g(c["gaps"]) if c["gaps"] else None, output_file=None, no_gc_correct=True,
                     remove_nocov=c["nocov"], merge_bins=c["merge"], quality_threshold=c.get("q", 30), workers=1)
        gz_write(g(f"delfi_driver_{name}.csv.gz"), frame_text(df))
        J["cases"][name] = dict(c, rows=int(df.shape[0]), columns=list(df.columns),
                                sha256=hashlib.sha256(frame_text(df).encode()).hexdigest())
        print(name, df.shape, list(df.columns))
    assert J["cases"]["bins20_gaps_bl"]["rows"] > 13_665 - 2, "the positional drop must bite"
    # the writers, on the merged case (small) and the `-` writer on it too
    wc = cases["bins20_merged"]
    outs = {}
    for suffix in (".tsv", ".bed", ".csv", ".bed.gz", "-", ".txt"):
        out = "-" if suffix == "-" else os.path.join(tmp, "out" + suffix)
        buf = io.StringIO()
        try:
            with contextlib.redirect_stdout(buf):
                import finaletoolkit.frag._delfi as RD
                RD.stdout = buf  # `from sys import stdout` was bound at import
                F.delfi(frag, g("synth.chrom.sizes"), g(wc["bins"]), fasta, blacklist_file=g("synth_blacklist.bed"),
                        gap_file=g(wc["gaps"]), output_file=out, no_gc_correct=True, merge_bins=True, workers=1)
            if suffix == "-":
                data = buf.getvalue().encode()
            else:
                data = open(out, "rb").read()
            keep = g("delfi_driver_out" + (".stdout" if suffix == "-" else suffix))
            open(keep, "wb").write(data)
            outs[suffix] = dict(ok=True, bytes=len(data), sha256=hashlib.sha256(data).hexdigest())
        except Exception as e:  # noqa: BLE001 - the error type is the golden
            outs[suffix] = dict(ok=False, error=type(e).__name__, message=str(e)[:200])
        print("writer", suffix, outs[suffix])
    J["writers"] = outs
    with open(g("delfi_driver.json"), "w") as fh:
        json.dump(J, fh, indent=1, sort_keys=True)
    print("delfi driver goldens written to", GOLD)


if __name__ == "__main__":
    main()
