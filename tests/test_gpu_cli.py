"""GPU: the command line end to end -- BASELINE config 1 (tiny BAM, coverage over ten windows) and the
next-row commands (adjust-wps, end-motifs, interval-end-motifs, mds, regional-mds)."""
import gzip
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = H.ROOT
DATA, GOLD = H.DATA, H.GOLDEN
BAM = os.path.join(DATA, "12.3444.b37.bam")


def cli(*args):
    r = subprocess.run([sys.executable, "-m", "finaletoolkit_amd.cli", *map(str, args)], capture_output=True,
                       text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def test_config1_cli_coverage_on_the_bam(tmp_path):
    """Ten 400 bp windows tiling 12:34443000-34447000 through `coverage` on the BAM fixture; the expected
    counts come from the oracle in BAM read1-fetch mode on the decoder's columns."""
    from oracle import oracle as O
    from tests.test_abi import _decode
    bed = tmp_path / "ten.bed"
    starts = list(range(34443000, 34447000, 400))
    bed.write_text("".join(f"12\t{s}\t{s + 400}\tw{i}\n" for i, s in enumerate(starts)))
    out = cli("coverage", BAM, bed, "-o", "-", "-q", 0)
    rows = [l.split("\t") for l in out.splitlines()]
    assert [r[3] for r in rows] == [f"w{i}" for i in range(10)]
    _, cols, _ = _decode(BAM, bam=True)["12"]
    fr = O.Frags(*cols)
    want = O.c_window_counts(fr, np.array(starts, np.int32), np.array(starts, np.int32) + 400, mapq_min=0)
    assert [float(r[4]) for r in rows] == [float(v) for v in want]
    assert sum(want) > 0


def test_cli_motif_commands(tmp_path):
    A = np.load(os.path.join(GOLD, "motifs.npz"))
    genome = H.read_fasta_gz(os.path.join(GOLD, "synth_ref.fa.gz"))
    fa = tmp_path / "ref.fa"
    H.write_fasta(fa, genome)
    frags = os.path.join(GOLD, "synth.frag.gz")
    tsv = tmp_path / "em.tsv"
    cli("end-motifs", frags, fa, "-k", 4, "-q", 30, "-o", tsv)
    got = np.array([float(l.split("\t")[1]) for l in open(tsv)])
    assert np.array_equal(got, A["end_k4_both"])
    mds = float(cli("mds", tsv).strip())
    assert mds == pytest.approx(json.load(open(os.path.join(GOLD, "motifs.json")))["end_k4_both"]["mds"], rel=1e-12)
    iv = tmp_path / "iv.tsv"
    cli("interval-end-motifs", frags, fa, os.path.join(GOLD, "motif_intervals.bed"), "-k", 4, "-q", 30, "-o", iv)
    lines = open(iv).read().splitlines()
    assert lines[0].startswith("contig\tstart\tstop\tname\tcount\tAAAA") and len(lines) == 1 + len(A["iv_end_k4_both"])
    assert [int(l.split("\t")[4]) for l in lines[1:]] == A["iv_end_k4_both"].sum(axis=1).tolist()
    rmds = tmp_path / "rmds.bed"
    cli("regional-mds", iv, rmds, "-s", "\t")
    vals = [float(l.split("\t")[4]) for l in open(rmds)]
    # the table stores 6-decimal frequencies, so the score is the golden one up to that rounding
    np.testing.assert_allclose(vals, A["iv_end_k4_both_mds"], rtol=0, atol=5e-4)
    bp = tmp_path / "bp.tsv"
    cli("breakpoint-motifs", frags, fa, "-k", 6, "-q", 30, "--strand", "both", "-o", bp)
    got = np.array([float(l.split("\t")[1]) for l in open(bp)])
    assert np.array_equal(got, A["bp_k6_both"])


def test_cli_adjust_wps(tmp_path):
    from finaletoolkit_amd.bigwig import BigWigFile, write_fixed_step_bigwig
    A = np.load(os.path.join(GOLD, "adjust_wps.npz"))
    cases = {c["key"]: c for c in json.load(open(os.path.join(GOLD, "adjust_wps.json")))}
    runs = {}
    for k in A.files:
        if k.startswith("track_") and k.endswith("_start"):
            _, c, i, _ = k.split("_")
            runs.setdefault(c, []).append((int(A[k]), A[f"track_{c}_{i}_values"]))
    sizes = os.path.join(GOLD, "adjust.chrom.sizes")
    header = [(l.split()[0], int(l.split()[1])) for l in open(sizes)]
    raw = tmp_path / "raw.bw"
    write_fixed_step_bigwig(str(raw), header, [(c, s0, v) for c, _ in header for s0, v in sorted(runs[c])])
    out = tmp_path / "adj.bw"
    cs = cases["w200_nosavgol"]
    cli("adjust-wps", raw, os.path.join(GOLD, "adjust_sites.bed"), sizes, "-o", out, "-i", 3000, "-m", 200, "--no-savgol")
    bw = BigWigFile(str(out))
    for i in range(cs["n_runs"]):
        want = A[f"w200_nosavgol_{i}_values"].astype(np.float32)
        s = int(A[f"w200_nosavgol_{i}_start"])
        st, en, v = bw.intervals(cs["run_contigs"][i], s, s + len(want))
        assert st[0] == s and np.array_equal(v.astype(np.float32), want)


def test_one_gpu_commands_run_without_torch(tmp_path):
    """A command on one GPU is a fresh process whose wall time is mostly start-up; ``import torch`` would add seconds
    to it and nothing on the path needs it (torch is the launcher and the exchange of the MULTI-rank commands).  Every
    sharded command's one-rank form is run here in one child process, which must end without torch loaded."""
    frag_file = os.path.join(DATA, "12.3444.b37.frag.gz")
    bed = os.path.join(DATA, "intervals.bed")
    code = f"""
import sys
from finaletoolkit_amd import frag
cov = frag.coverage({frag_file!r}, {bed!r}, {str(tmp_path / 'c.bed')!r}, normalize=True)
assert len(cov) == 2 and cov[0].coverage > 0
st = frag.frag_length_intervals({frag_file!r}, {bed!r}, None)
assert len(st) == 2
bins = frag.frag_length_bins({frag_file!r}, '12', 34443000, 34447000, bin_size=5)
w = frag.wps({frag_file!r}, '12', 34443000, 34447000, 133851895, output_file={str(tmp_path / 'w.wig')!r})
assert len(w) == 4000
frag.multi_wps({frag_file!r}, {bed!r}, {os.path.join(DATA, 'b37.chrom.sizes')!r}, {str(tmp_path / 'm.bed.gz')!r}, interval_size=400)
assert 'torch' not in sys.modules, sorted(m for m in sys.modules if m.startswith('torch'))[:5]
print('ok')
"""
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stderr[-2000:]
