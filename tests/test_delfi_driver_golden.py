"""The DELFI DRIVER pinned to the reference: ``frag.delfi`` against frames and files the imported reference's own
``delfi(..., workers=1)`` produced (``oracle/gen_golden_delfi.py``; reference ``frag/_delfi.py:129-401``): gap-overlap
bin filter, chrom.sizes order (the bins file lists chrB first), NOARM drop, ratio, the positional 8779 / 13664 drop
with ``reset_index`` on > 13 665 surviving bins, short-arm contigs, bins beyond the contig end (GC skipped), a bins
contig chrom.sizes lacks, ``delfi_merge_bins`` and every writer of ``_write_delfi`` (``.bed.gz`` raises LookupError
in the reference -- pandas is handed ``encoding="gzip"`` -- and so does the product).

CPU: the driver's host logic with the device replaced by the oracle (counts) and a host GC count -- no GPU.
GPU (``-m gpu``): the real ``frag.delfi`` (HIP counts, device GC)."""
import contextlib
import gzip
import io
import json
import os
import warnings

import numpy as np
import pytest

from tests import helpers as H

GOLD = H.GOLDEN
CONTIGS = {"chrA": 400_000, "chrB": 150_000}
J = json.load(open(os.path.join(GOLD, "delfi_driver.json")))


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    d = tmp_path_factory.mktemp("delfi_driver")
    fasta = str(d / "synth_ref.fa")
    assert H.synth_reference(fasta, CONTIGS) == J["fasta_sha256"]  # the file the reference ran on
    H.write_bins20(str(d / "synth_bins20.bed"), CONTIGS)
    return d, fasta


def _path(d, name):
    return str(d / name) if name == "synth_bins20.bed" else os.path.join(GOLD, name)


def _golden_text(name):
    return gzip.open(os.path.join(GOLD, f"delfi_driver_{name}.csv.gz"), "rt").read()


def _run(delfi, d, fasta, c, output_file=None):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return delfi(os.path.join(GOLD, "synth.frag.gz"), os.path.join(GOLD, c.get("sizes", "synth.chrom.sizes")), _path(d, c["bins"]),
                     fasta, blacklist_file=os.path.join(GOLD, "synth_blacklist.bed") if c["bl"] else None,
                     gap_file=_path(d, c["gaps"]) if c["gaps"] else None, output_file=output_file, no_gc_correct=True,
                     remove_nocov=c["nocov"], merge_bins=c["merge"], quality_threshold=c.get("q", 30), workers=1)


def _check_frame(df, name):
    c = J["cases"][name]
    assert list(df.columns) == c["columns"] and df.shape[0] == c["rows"], name
    assert df.to_csv(index=False) == _golden_text(name), name


def _check_writers(delfi, d, fasta, tmp_path):
    wc = dict(J["cases"]["bins20_merged"])
    for suffix, want in J["writers"].items():
        out = "-" if suffix == "-" else str(tmp_path / ("out" + suffix))
        if not want["ok"]:
            with pytest.raises({"LookupError": LookupError, "ValueError": ValueError}[want["error"]]):
                _run(delfi, d, fasta, wc, out)
            continue
        if suffix == "-":
            import finaletoolkit_amd.frag._delfi as D
            buf, keep = io.StringIO(), D.stdout
            D.stdout = buf
            try:
                with contextlib.redirect_stdout(buf):
                    _run(delfi, d, fasta, wc, out)
            finally:
                D.stdout = keep
            got = buf.getvalue().encode()
            ref = open(os.path.join(GOLD, "delfi_driver_out.stdout"), "rb").read()
        else:
            _run(delfi, d, fasta, wc, out)
            got = open(out, "rb").read()
            ref = open(os.path.join(GOLD, "delfi_driver_out" + suffix), "rb").read()
        assert got == ref, suffix


# ---- CPU: the driver with an oracle-backed device --------------------------------------------------------------
@pytest.fixture()
def host_delfi(monkeypatch):
    from finaletoolkit_amd.frag import _delfi as D
    from oracle import oracle as O
    cols = H.read_frag_gz(os.path.join(GOLD, "synth.frag.gz"))
    frs = {c: O.Frags(*v) for c, v in cols.items()}

    class Src:
        contigs = list(cols)
        lengths = {c: None for c in cols}

        def load_all(self):
            pass

        def has(self, c):
            return c in cols

        def require(self, c):
            return c

        def require_interval(self, c, *a, **k):
            return self.require(c)

    class Eng:
        def delfi_counts(self, name, starts, stops, q=30, bs=None, be=None, gaps=None):
            return O.c_delfi_counts(frs[name], starts, stops, q, bs, be, gaps)

    class Ref:
        def __init__(self, path):
            self.seqs = {}
            name = None
            for line in open(path):
                if line.startswith(">"):
                    name = line[1:].split()[0]
                    self.seqs[name] = []
                else:
                    self.seqs[name].append(line.strip())
            self.seqs = {k: "".join(v).upper() for k, v in self.seqs.items()}
            self.chroms = {k: len(v) for k, v in self.seqs.items()}

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            pass

        def gc_counts(self, eng, contig, starts, stops):
            s = self.seqs[contig]
            return np.array([s[a:b].count("G") + s[a:b].count("C") for a, b in zip(starts, stops)], np.int64)

    monkeypatch.setattr(D, "resident_contigs", lambda path, names, *a, **k: ((Src(), c) for c in names if c in cols))
    monkeypatch.setattr(D, "get_engine", lambda: Eng())
    monkeypatch.setattr(D, "ReferenceGenome", Ref)
    return D.delfi


@pytest.mark.parametrize("name", sorted(J["cases"]))
def test_driver_frames_equal_the_reference_host_logic(host_delfi, inputs, name):
    d, fasta = inputs
    _check_frame(_run(host_delfi, d, fasta, J["cases"][name]), name)


def test_driver_writers_equal_the_reference_host_logic(host_delfi, inputs, tmp_path):
    d, fasta = inputs
    _check_writers(host_delfi, d, fasta, tmp_path)


# ---- GPU: the product -----------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(J["cases"]))
def test_driver_frames_equal_the_reference(inputs, name):
    from finaletoolkit_amd import frag
    d, fasta = inputs
    _check_frame(_run(frag.delfi, d, fasta, J["cases"][name]), name)


@pytest.mark.gpu
def test_driver_writers_equal_the_reference(inputs, tmp_path):
    from finaletoolkit_amd import frag
    d, fasta = inputs
    _check_writers(frag.delfi, d, fasta, tmp_path)


@pytest.mark.gpu
def test_driver_with_a_2bit_reference_gives_the_same_frames(inputs, tmp_path):
    """The same cases with the reference genome as .2bit (py2bit in the reference): identical frames."""
    from finaletoolkit_amd import frag
    d, fasta = inputs
    seqs, name = {}, None
    for line in open(fasta):
        if line.startswith(">"):
            name = line[1:].split()[0]
            seqs[name] = []
        else:
            seqs[name].append(line.strip())
    H.write_2bit(tmp_path / "ref.2bit", {k: "".join(v) for k, v in seqs.items()})
    for name in ("bins20_gaps_bl", "mixed_gaps", "bins20_merged"):
        _check_frame(_run(frag.delfi, d, str(tmp_path / "ref.2bit"), J["cases"][name]), name)
