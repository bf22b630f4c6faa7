"""
finaletoolkit_amd -- MI355X-native engine for FinaleToolkit's per-window hot
path (coverage, WPS, fragment-length histograms, DELFI short/long bins).

Host Python keeps the ``finaletoolkit.frag.*`` surface
(``finaletoolkit_amd.frag``); the per-fragment loops run as HIP kernels behind
the C ABI of ``include/ftk.h`` (``libftk_hip.so``).  No CPU fallback.
"""
import importlib as _importlib
import importlib.util as _importlib_util
import os as _os
import sys as _sys


def _hardware_queues():
    """The ONE import side effect, and how to turn it off.  The streaming decoder keeps up to eight pieces of a file in
    flight on HIP streams of their own; the HIP runtime multiplexes a process's streams onto ``GPU_MAX_HW_QUEUES``
    hardware queues (ROCm's default: 4) and two streams on one queue run strictly one after the other: whole-genome
    DELFI leg 0.176-0.183 s with 4 queues, 0.151-0.162 s with 16 (tools/e2e_genome_bench.py).  The runtime reads the
    variable ONCE, at its first HIP call - which in most scripts is torch's, long before the engine is first used
    (round 4 measured exactly that: set at library load instead, a script that touched ``torch.cuda`` first ran the leg
    18 % slower) - so it is set here, when the package is imported.  It is process-wide (torch / RCCL in this process
    and child processes see it).  A value the user has set wins; ``FTK_HW_QUEUES=<n>`` chooses another number and
    ``FTK_HW_QUEUES=0`` leaves ROCm's default alone."""
    want = _os.environ.get("FTK_HW_QUEUES", "16").strip()
    if want in ("", "0", "off", "default") or "GPU_MAX_HW_QUEUES" in _os.environ:
        return
    if not want.isdigit():
        raise ValueError(f"FTK_HW_QUEUES={want!r}: expected a number of hardware queues, or 0 to keep ROCm's default")
    _os.environ["GPU_MAX_HW_QUEUES"] = want


_hardware_queues()

from .exceptions import (FinaleToolkitError, InvalidInputError, MissingIndexError,  # noqa: F401
                         UnsupportedFormatError)

__version__ = "0.1.0"

# Flat namespace of the reference (``finaletoolkit.<name>``, src/finaletoolkit/__init__.py:49-128), resolved on first
# use (PEP 562) so that importing the package for its names loads neither pandas nor the HIP library.  Listed per
# submodule: every hot-path export that exists here.  Reference exports outside the path (``filter_file``,
# ``frag_bam_to_bed``, ``low_quality_read_pairs``, ``ReferenceWrapper``) are named in
# ``_OUT_OF_SCOPE`` so that asking for one says why it is absent.
_SUBMODULES = ("cli", "frag", "genome", "io", "utils")
_FLAT_BY_MODULE = {
    "frag": ("frag_length", "frag_length_bins", "frag_length_intervals", "coverage", "single_coverage", "wps",
             "multi_wps", "adjust_wps", "cleavage_profile", "multi_cleavage_profile", "delfi", "delfi_gc_correct",
             "delfi_merge_bins", "end_motifs", "region_end_motifs", "interval_end_motifs", "EndMotifFreqs",
             "EndMotifsIntervals", "breakpoint_motifs", "region_breakpoint_motifs", "interval_breakpoint_motifs",
             "BreakpointMotifFreqs", "BreakpointMotifsIntervals", "CoverageResult", "FragLengthStats"),
    "utils": ("frag_generator", "frag_array", "frags_in_region", "agg_bw", "get_intervals", "overlaps", "gen_kmers",
              "chrom_sizes_to_dict", "chrom_sizes_to_list", "reverse_complement"),
    "genome": ("GenomeGaps", "ContigGaps", "ucsc_hg19_gap_bed", "b37_gap_bed", "ucsc_hg38_gap_bed"),
    "io": ("Fragment", "AlignmentWrapper"),
}
_FLAT = {name: module for module, names in _FLAT_BY_MODULE.items() for name in names}
_SINGULAR = {"end_motif": "end_motifs", "breakpoint_motif": "breakpoint_motifs"}
_OUT_OF_SCOPE = ("filter_file", "frag_bam_to_bed", "low_quality_read_pairs", "ReferenceWrapper")


def __getattr__(name):
    if name in _SUBMODULES:
        return _importlib.import_module("." + name, __name__)
    module = _FLAT.get(_SINGULAR.get(name, name))
    if module is not None:
        value = getattr(_importlib.import_module("." + module, __name__), _SINGULAR.get(name, name))
        globals()[name] = value
        return value
    if name in _OUT_OF_SCOPE:
        raise AttributeError(f"{__name__}.{name}: this reference utility is outside the accelerated hot path "
                             "(SURVEY.md section 8) and is not provided; use the reference package for it")
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def __dir__():
    return sorted(set(globals()) | set(_SUBMODULES) | set(_FLAT) | set(_SINGULAR))


def install_alias(name: str = "finaletoolkit", force: bool = False):
    """Make ``import finaletoolkit`` (and ``finaletoolkit.frag`` / ``.utils`` / ``.genome`` / ``.io`` / ``.cli`` /
    ``.exceptions``) resolve to this package, so a script written against the reference runs unchanged:

        import finaletoolkit_amd; finaletoolkit_amd.install_alias()
        import finaletoolkit as ft; ft.frag.delfi(...); ft.coverage(...)
        from finaletoolkit.frag import wps

    Opt-in, per process.  Refuses (``ImportError``) when a real ``finaletoolkit`` distribution is importable, unless
    ``force`` - two packages answering to one name is the caller's decision.  Returns the package."""
    this = _sys.modules[__name__]
    have = _sys.modules.get(name)
    if have is not None and have is not this and not force:
        raise ImportError(f"a different module is already imported as {name!r}; pass force=True to shadow it")
    if have is None and not force:
        try:
            found = _importlib_util.find_spec(name)
        except (ImportError, ValueError):
            found = None
        if found is not None:
            raise ImportError(f"{name!r} is installed ({found.origin}); pass force=True to shadow it in this process")
    _sys.modules[name] = this
    for sub in _SUBMODULES + ("exceptions",):
        module = _importlib.import_module("." + sub, __name__)
        _sys.modules[f"{name}.{sub}"] = module
    # the reference keeps its utilities in a package; ``finaletoolkit.utils.utils`` and ``finaletoolkit.utils.validation``
    # are the deep paths scripts (and the reference's own tests) use
    utils = _sys.modules[f"{name}.utils"]
    _sys.modules[f"{name}.utils.utils"] = utils
    utils.utils = utils
    validation = _importlib.import_module(".validation", __name__)
    _sys.modules[f"{name}.utils.validation"] = validation
    utils.validation = validation
    return this
