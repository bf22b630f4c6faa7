"""
Windowed Protection Score over one interval -- ``wps`` with the reference's
signature, result dtype, WIG writer, deprecation handling and degenerate-
interval behaviour (``src/finaletoolkit/frag/_wps.py:56-229``).  The per-base
loop (``:180-188``) is the ``ftk_wps`` kernel.
"""
from __future__ import annotations

import time
import warnings
from pathlib import Path
from sys import stderr, stdout
from typing import Union

import numpy as np

from .. import sharding
from ..source import get_engine, open_source

__all__ = ["wps"]

_WPS_DTYPE = [("contig", "U16"), ("start", "i8"), ("wps", "i8")]


def _resolve_aliases(min_length, max_length, fraction_low, fraction_high):
    """The deprecated spellings of the length bounds (the reference's frag/_wps.py:112-140, frag/_multi_wps.py:105-133):
    an old name warns and stands in for the new one; giving both is an error (after the warning)."""
    bounds = []
    for new, value, old, alias in (("min_length", min_length, "fraction_low", fraction_low),
                                   ("max_length", max_length, "fraction_high", fraction_high)):
        if alias is not None:
            warnings.warn(f"{old} is deprecated. Use {new} instead.", category=DeprecationWarning, stacklevel=3)
            if value is not None:
                raise ValueError(f"{old} and {new} cannot both be specified")
            value = alias
        bounds.append(value)
    return bounds[0], bounds[1]


def _scores_array(chrom, start, values):
    """The reference's result array (frag/_wps.py:181-188): 80-byte records, gigabytes for a chromosome -- large
    ones are filled by the library's host threads (``ftk_fill_wps_records``), small ones by numpy."""
    n = len(values)
    if n >= 1 << 20:
        from .. import _lib as L
        scores = np.empty(n, dtype=_WPS_DTYPE)
        name = np.zeros(1, dtype="U16")
        name[0] = chrom  # numpy's own truncation / padding of the name
        vals = np.ascontiguousarray(values, dtype=np.int64)
        rc = L.load().ftk_fill_wps_records(L.ptr(scores), n, L.ptr(name.view(np.uint32)), int(start), L.ptr(vals), 0)
        if rc == L.FTK_OK:
            return scores
    scores = np.zeros(n, dtype=_WPS_DTYPE)
    scores["contig"] = chrom
    scores["start"] = np.arange(start, start + n, dtype=np.int64)
    scores["wps"] = values
    return scores


def wps(input_file: Union[str, Path], chrom: str, start: int, stop: int, chrom_size: int,
        output_file: str | None = None, window_size: int = 120, min_length: int = 120, max_length: int = 180,
        quality_threshold: int = 30, verbose: bool | int = 0, fraction_low: int | None = None,
        fraction_high: int | None = None, reference_file: str | Path | None = None) -> np.ndarray:
    """Raw WPS for every base of ``chrom:[start, stop)``; structured array with
    fields ``('contig', 'start', 'wps')``."""
    if verbose:
        t0 = time.time()
        stderr.write(f"[finaletoolkit-wps] Region: {chrom}:{start}-{stop}\n")
    min_length, max_length = _resolve_aliases(min_length, max_length, fraction_low, fraction_high)
    start = int(start)
    stop = int(stop)
    if stop <= start:
        warnings.warn(f"[wps] {chrom}:{start}-{stop} is a degenerate interval (stop <= start); skipping.",
                      UserWarning, stacklevel=2)
        return np.zeros(0, dtype=_WPS_DTYPE)
    src = open_source(input_file)
    eng = get_engine()
    # rows the call can need: tabix rows overlapping a base's window (window_size / 2 either side); for a BAM the
    # read1 alignments overlapping the reference's fetch window [start - max_length, stop + max_length)
    # (frag/_wps.py:156-157) - a read1 up to max_length outside the interval still brings its fragment in
    pad = max(int(window_size), int(max_length)) + 1
    values = eng.wps(src.require_interval(chrom, start, stop, pad), start, stop, int(chrom_size), int(window_size),
                     0 if min_length is None else int(min_length), int(max_length), int(quality_threshold))
    scores = _scores_array(chrom, start, values)

    if isinstance(output_file, str):
        if not (output_file.endswith((".wig.gz", ".wig")) or output_file == "-"):
            raise ValueError("output_file can only have suffixes .wig or .wig.gz.")
        if sharding.is_writer():  # one region is not sharded: under several ranks rank 0 alone writes it
            # (the engine's contiguous scores, not the 80-byte-strided field of the record array: gathering that
            # field back cost 0.125 s for chr22, more than formatting and writing the file)
            _write_wig(output_file, chrom, start, stop, values)
    elif output_file is not None:
        raise TypeError(f'output_file is unsupported type "{type(input_file)}". output_file should be a string '
                        "specifying the path of the file to output scores to.")
    if verbose:
        stderr.write(f"wps took {time.time() - t0} s to complete\n")
    return scores


def _write_wig(output_file, chrom, start, stop, values) -> None:
    """fixedStep WIG (frag/_wps.py:208-229): the header line, then one score per line -- the lines are
    formatted by the library's host threads (``writers.wig_body``), ``.wig.gz`` as parallel gzip members."""
    from .. import writers
    header = f"fixedStep\tchrom={chrom}\tstart={start}\tstep={1}\tspan={stop - start}\n"
    if not (output_file.endswith((".wig.gz", ".wig")) or output_file == "-"):
        raise ValueError("output_file can only have suffixes .wig or .wig.gz.")
    with writers.wig_body(values) as body:
        if output_file == "-":
            stdout.write(header)
            stdout.write(body.tobytes().decode())
            stdout.flush()
            return
        level = writers.GZIP_LEVEL if output_file.endswith(".wig.gz") else 0
        writers.write_text(output_file, header.encode(), level)
        body.write(output_file, level, append=True)
