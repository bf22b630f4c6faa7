"""
The reference's import names (SURVEY §8-b upper side; reference ``src/finaletoolkit/__init__.py:49-128``):
``finaletoolkit_amd.<name>`` resolves lazily, and after the opt-in ``install_alias()`` a script written against the
reference - ``import finaletoolkit as ft``, ``ft.frag.delfi``, ``ft.coverage``, ``from finaletoolkit.frag import wps`` -
runs unchanged.  The CPU tests cover the name table; the GPU tests run the reference's own call spellings on the
committed fixtures against the reference-generated goldens.
"""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.helpers import DATA, ROOT, golden_json

# the reference's flat exports (its __init__.py:_EXPORTS keys), split by whether the hot path covers them
ON_PATH = ["frag_length", "frag_length_bins", "frag_length_intervals", "coverage", "single_coverage", "wps", "multi_wps",
           "adjust_wps", "cleavage_profile", "multi_cleavage_profile", "delfi", "delfi_gc_correct", "delfi_merge_bins",
           "end_motifs", "region_end_motifs", "interval_end_motifs", "EndMotifFreqs", "EndMotifsIntervals",
           "breakpoint_motifs", "region_breakpoint_motifs", "interval_breakpoint_motifs", "BreakpointMotifFreqs",
           "BreakpointMotifsIntervals", "frag_generator", "frag_array", "frags_in_region", "agg_bw", "get_intervals",
           "overlaps", "gen_kmers", "reverse_complement", "chrom_sizes_to_dict", "chrom_sizes_to_list", "GenomeGaps", "ContigGaps",
           "ucsc_hg19_gap_bed", "b37_gap_bed", "ucsc_hg38_gap_bed", "Fragment", "AlignmentWrapper", "end_motif", "breakpoint_motif"]
OFF_PATH = ["filter_file", "frag_bam_to_bed", "low_quality_read_pairs", "ReferenceWrapper"]


@pytest.fixture()
def alias():
    import finaletoolkit_amd
    before = {k: v for k, v in sys.modules.items() if k == "finaletoolkit" or k.startswith("finaletoolkit.")}
    finaletoolkit_amd.install_alias()
    yield finaletoolkit_amd
    for k in [k for k in sys.modules if k == "finaletoolkit" or k.startswith("finaletoolkit.")]:
        del sys.modules[k]
    sys.modules.update(before)


def test_import_is_lazy_and_its_one_side_effect_can_be_turned_off():
    code = ("import os, sys; sys.path.insert(0, %r); import finaletoolkit_amd as f; "
            "print(os.environ.get('GPU_MAX_HW_QUEUES', '-')); "
            "assert 'pandas' not in sys.modules and 'finaletoolkit_amd.frag' not in sys.modules; "
            "assert 'torch' not in sys.modules and 'finaletoolkit' not in sys.modules; "
            "f.get_intervals; assert 'finaletoolkit_amd.utils' in sys.modules; print('ok')" % ROOT)
    base = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "FTK_HW_QUEUES")}
    for extra, want in (({}, "16"), ({"FTK_HW_QUEUES": "0"}, "-"), ({"FTK_HW_QUEUES": "8"}, "8"),
                        ({"GPU_MAX_HW_QUEUES": "2"}, "2"), ({"GPU_MAX_HW_QUEUES": "2", "FTK_HW_QUEUES": "16"}, "2")):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(base, **extra))
        assert out.returncode == 0 and out.stdout.split() == [want, "ok"], (extra, out.stdout, out.stderr)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(base, FTK_HW_QUEUES="many"))
    assert out.returncode != 0 and "FTK_HW_QUEUES" in out.stderr


def test_flat_names_resolve_to_the_submodule_objects():
    import finaletoolkit_amd as f
    for name in ON_PATH:
        assert getattr(f, name) is not None, name
        assert name in dir(f)
    assert f.coverage is f.frag.coverage and f.wps is f.frag.wps and f.delfi is f.frag.delfi
    assert f.frag_generator is f.utils.frag_generator and f.GenomeGaps is f.genome.GenomeGaps
    assert f.end_motif is f.frag.end_motifs and f.breakpoint_motif is f.frag.breakpoint_motifs
    frag = f.Fragment("12", 10, 177, 60, True)
    assert frag.length == 167 and frag == ("12", 10, 177, 60, True)
    for name in OFF_PATH:
        with pytest.raises(AttributeError, match="outside the accelerated hot path"):
            getattr(f, name)
    with pytest.raises(AttributeError):
        f.no_such_name


def test_alias_gives_the_reference_spellings(alias):
    import finaletoolkit as ft
    import finaletoolkit.frag
    import finaletoolkit.utils.utils as deep
    from finaletoolkit.exceptions import InvalidInputError
    from finaletoolkit.frag import delfi_merge_bins, wps
    from finaletoolkit.genome import GenomeGaps
    from finaletoolkit.utils import chrom_sizes_to_list, frag_generator, get_intervals
    assert ft is alias and finaletoolkit.frag is alias.frag
    assert wps is alias.frag.wps and ft.coverage is alias.frag.coverage and ft.frag.delfi is alias.frag.delfi
    assert deep.get_intervals is get_intervals and frag_generator is alias.utils.frag_generator
    assert issubclass(InvalidInputError, ValueError) and GenomeGaps is alias.genome.GenomeGaps
    assert get_intervals(os.path.join(DATA, "intervals.bed"))[0][0] == "12"
    assert chrom_sizes_to_list(os.path.join(DATA, "b37.chrom.sizes"))[0] == ("1", 249250621)
    assert callable(delfi_merge_bins)
    assert importlib.import_module("finaletoolkit.frag") is alias.frag


def test_alias_refuses_to_shadow_silently(tmp_path, monkeypatch):
    import finaletoolkit_amd
    pkg = tmp_path / "finaletoolkit"
    pkg.mkdir()
    (pkg / "__init__.py").write_text("REAL = True\n")
    monkeypatch.syspath_prepend(str(tmp_path))
    importlib.invalidate_caches()
    assert "finaletoolkit" not in sys.modules
    with pytest.raises(ImportError, match="is installed"):
        finaletoolkit_amd.install_alias()
    import finaletoolkit
    assert finaletoolkit.REAL
    with pytest.raises(ImportError, match="already imported"):
        finaletoolkit_amd.install_alias()
    try:
        assert finaletoolkit_amd.install_alias(force=True) is finaletoolkit_amd
        assert sys.modules["finaletoolkit"] is finaletoolkit_amd
    finally:
        for k in [k for k in sys.modules if k == "finaletoolkit" or k.startswith("finaletoolkit.")]:
            del sys.modules[k]


@pytest.mark.gpu
def test_a_reference_script_runs_unchanged(alias, tmp_path):
    """The calls of the reference's tests/test_coverage.py:15-89, tests/test_wps.py:18-26 and
    tests/test_frag_io.py:15-109, spelled as a user of the reference spells them."""
    import finaletoolkit as ft
    from finaletoolkit.frag import single_coverage, wps
    from finaletoolkit.utils import frag_generator
    G = golden_json()
    fix = os.path.join(DATA, "12.3444.b37.frag.gz")
    bam = os.path.join(DATA, "12.3444.b37.bam")
    assert single_coverage(bam, "12", 34443400, 34443600, quality_threshold=0).coverage == 2
    assert ft.single_coverage(bam, "12", 34443000, 34447000, quality_threshold=0).coverage == 17
    c = G["fixture"]["single_coverage"][0]
    assert ft.frag.single_coverage(fix, "12", c["start"], c["stop"], quality_threshold=c["q"],
                                   intersect_policy=c["policy"], min_length=c["min_length"],
                                   max_length=c["max_length"]).coverage == c["coverage"]
    res = ft.coverage(fix, os.path.join(DATA, "intervals.bed"), None, normalize=False, intersect_policy="midpoint",
                      scale_factor=1.)
    assert [list(r) for r in res] == G["fixture"]["coverage_raw"]
    ft.coverage(fix, os.path.join(DATA, "intervals.bed"), str(tmp_path / "c.bed"), normalize=True, scale_factor=1e6)
    assert open(tmp_path / "c.bed").read() == G["fixture"]["coverage_norm_bed_text"]
    scores = wps(fix, "12", 34444145, 34444155, chrom_size=133851895, quality_threshold=0)
    assert scores["wps"].tolist() == [-1] * 5 + [1] * 5
    assert np.array_equal(ft.wps(bam, "12", 34444145, 34444155, chrom_size=133851895, quality_threshold=0)["wps"],
                          scores["wps"])
    frags = list(frag_generator(fix, "12", quality_threshold=0, min_length=0, max_length=9999))
    assert [list(f) for f in frags] == G["fixture"]["frag_generator_all"]
    stats = ft.frag_length_intervals(fix, os.path.join(DATA, "intervals.bed"))
    assert len(stats) == 2 and stats[1].median == 147.0  # the reference's odd-count median quirk


@pytest.mark.gpu
def test_alias_in_a_fresh_process_cli_module(tmp_path):
    """``python -m finaletoolkit.cli`` is out of reach of an in-process alias; the documented spelling is
    ``python -m finaletoolkit_amd.cli``.  A fresh process with the alias installed before the user's imports."""
    code = ("import sys; sys.path.insert(0, %r); import finaletoolkit_amd; finaletoolkit_amd.install_alias(); "
            "import finaletoolkit as ft; "
            "print(int(ft.single_coverage(%r, '12', 34443400, 34443600, quality_threshold=0)[4]))"
            % (ROOT, os.path.join(DATA, "12.3444.b37.frag.gz")))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().splitlines()[-1] == "2", out.stderr


def test_validation_helpers_keep_the_references_contract(alias):
    """``utils/validation.py`` of the reference, by the cases of its own ``tests/test_validation.py`` (29 of them:
    return values, exception types, the words its tests match in the messages), imported the way it imports them."""
    from finaletoolkit.utils.validation import valid_interval, validate_compatible_contigs
    import finaletoolkit.utils as U
    assert U.valid_interval is valid_interval and U.validate_compatible_contigs is validate_compatible_contigs
    v = validate_compatible_contigs
    assert v(["1", "2", "3"], ["1", "2", "3"]) and v(["1", "2", "3"], ["1", "2"])
    with pytest.raises(ValueError, match="not found in reference"):
        v(["1", "2"], ["1", "2", "4"])
    assert not v(["1", "2"], ["1", "2", "4"], throw_on_error=False)
    with pytest.raises(ValueError, match="not found in input"):
        v(["1", "2", "3"], ["1", "2"], allow_subset=False)
    assert not v(["1", "2", "3"], ["1", "2"], allow_subset=False, throw_on_error=False)
    assert v(["1", "2"], ["1", "2"], allow_subset=False)
    assert v({"1": 100, "2": 200}, {"1": 100, "2": 200}, validate_sizes=True)
    with pytest.raises(RuntimeError, match="length mismatch"):
        v({"1": 100, "2": 200}, {"1": 100, "2": 999}, validate_sizes=True)
    assert not v({"1": 100, "2": 200}, {"1": 100, "2": 999}, validate_sizes=True, throw_on_error=False)
    with pytest.raises(TypeError, match="requires both"):
        v(["1", "2"], ["1", "2"], validate_sizes=True)
    assert not v(["1", "2"], ["1", "2"], validate_sizes=True, throw_on_error=False)
    assert v({"1": 100, "2": 999999}, {"1": 100}, validate_sizes=True)  # (only the input's contigs are measured)
    i = valid_interval
    assert not i(["1", "2"], "3")
    with pytest.raises(ValueError, match="not found in reference"):
        i(["1", "2"], "3", throw_on_error=True)
    assert i({"1": 1000}, "1", start=0, stop=1000)
    assert not i({"1": 1000}, "1", start=-1) and not i({"1": 1000}, "1", start=1000) and not i({"1": 1000}, "1", stop=-1)
    with pytest.raises(IndexError, match="out of bounds"):
        i({"1": 1000}, "1", start=-1, throw_on_error=True)
    with pytest.raises(IndexError, match="out of bounds"):
        i({"1": 1000}, "1", stop=1001, throw_on_error=True)
    assert not i({"1": 1000}, "1", start=500, stop=500)
    with pytest.raises(ValueError, match="must be less than stop"):
        i({"1": 1000}, "1", start=500, stop=100, throw_on_error=True)
    assert i({"1": 1000}, "1", start=999) and i({"1": 1000}, "1", stop=1000)
    assert i(["1", "2"], "1", start=10 ** 9) and i(["1", "2"], "1") and not i(["1", "2"], "1", start=-1)
    with pytest.raises(IndexError, match="cannot be negative"):
        i(["1", "2"], "1", start=-1, throw_on_error=True)


def test_small_pure_helpers():
    """``reverse_complement`` (utils/utils.py:413-437: either case in, upper case out, anything else kept) and the
    None-tolerant comparisons (utils/_comparison.py)."""
    import finaletoolkit_amd as f
    from finaletoolkit_amd.utils import _none_eq, _none_geq, _none_leq, reverse_complement
    assert f.reverse_complement is reverse_complement
    assert reverse_complement("ACGTN") == "NACGT" and reverse_complement("acgtn") == "nACGT" and reverse_complement("") == ""
    assert reverse_complement("AAC-x") == "x-GTT"
    assert _none_leq(None, 3) and _none_leq(3, None) and _none_leq(2, 3) and not _none_leq(4, 3)
    assert _none_geq(None, 3) and _none_geq(3, None) and _none_geq(3, 3) and not _none_geq(2, 3)
    assert _none_eq(None, 3) and _none_eq(3, None) and _none_eq(3, 3) and not _none_eq(2, 3)
