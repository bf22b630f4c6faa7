"""
``finaletoolkit.frag`` surface of the hot path, MI355X-backed (names, arguments
and results as in the reference's ``frag/__init__.py:7-34``).
"""
from ._adjust_wps import adjust_wps
from ._breakpoint_motifs import (BreakpointMotifFreqs, BreakpointMotifsIntervals, breakpoint_motifs,
                                 interval_breakpoint_motifs, region_breakpoint_motifs)
from ._cleavage_profile import cleavage_profile, multi_cleavage_profile
from ._coverage import CoverageResult, coverage, single_coverage
from ._delfi import delfi
from ._delfi_gc_correct import delfi_gc_correct
from ._delfi_merge_bins import delfi_merge_bins
from ._end_motifs import EndMotifFreqs, EndMotifsIntervals, end_motifs, interval_end_motifs, region_end_motifs
from ._frag_length import FragLengthStats, frag_length, frag_length_bins, frag_length_intervals
from ._multi_wps import multi_wps
from ._wps import wps

__all__ = ["frag_length", "frag_length_bins", "frag_length_intervals", "FragLengthStats", "coverage",
           "single_coverage", "CoverageResult", "wps", "multi_wps", "delfi", "delfi_gc_correct", "delfi_merge_bins",
           "cleavage_profile", "multi_cleavage_profile", "adjust_wps",
           "EndMotifFreqs", "EndMotifsIntervals", "end_motifs", "interval_end_motifs", "region_end_motifs",
           "BreakpointMotifFreqs", "BreakpointMotifsIntervals", "breakpoint_motifs", "interval_breakpoint_motifs",
           "region_breakpoint_motifs"]
