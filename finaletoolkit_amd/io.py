"""
Record type of the hot path (§8 a1; reference ``io/alignment.py:25-54``).

On the device a fragment is a row of the start-sorted SoA columns; this tuple is what crosses the Python
surface when a caller asks for fragments one by one (``utils.frag_generator``).  The reference's pysam-backed wrappers
(``AlignmentWrapper``, ``ReferenceWrapper``) have no counterpart: files are read by ``libftk_hip.so``'s own decoders
(``source.py``, ``reference.py``).
"""
from __future__ import annotations

from typing import NamedTuple

__all__ = ["Fragment"]


class Fragment(NamedTuple):
    """``(contig, start, stop, mapq, is_forward)``, 0-based half-open; compares equal to the plain 5-tuple."""
    contig: str
    start: int
    stop: int
    mapq: int
    is_forward: bool

    @property
    def length(self) -> int:
        return self.stop - self.start
