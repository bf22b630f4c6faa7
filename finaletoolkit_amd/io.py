"""
Record type and reader facade of the hot path (§8 a1-a3; reference ``io/alignment.py:25-302``).

On the device a fragment is a row of the start-sorted SoA columns; ``Fragment`` is what crosses the Python
surface when a caller asks for fragments one by one.  ``AlignmentWrapper`` keeps the reference's spelling for such a
caller - ``AlignmentWrapper(path, quality_threshold=...).fetch(contig, start, stop)``, the form the reference's own
drivers use (frag/_delfi.py:74-78,443) - over this package's decoders: the file is read by ``libftk_hip.so``
(``source.py``: BGZF inflate and row / record parsing on the GPU), a region query is an index read plus
``ftk_frag_select`` with the bare fetch rule (tabix: rows overlapping the region, io/alignment.py:270-302; BAM: read1
alignments overlapping it, flag-filtered, io/alignment.py:60-71,242-268), rows come back in file order.  There is
no pysam here: open pysam handles, CRAM and SAM are refused with ``UnsupportedFormatError``.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Generator, NamedTuple, Optional

__all__ = ["Fragment", "AlignmentWrapper"]


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


class AlignmentWrapper:
    """Fragments of a BAM or a tabix-indexed fragment file (``frag.gz`` / BED6 ``bed.gz``) by region.

    Same constructor arguments, attributes, exceptions (``FileNotFoundError``, ``MissingIndexError``,
    ``UnsupportedFormatError``) and BED6 warning as the reference's class (io/alignment.py:74-203).  ``threads`` are
    decoder threads.  ``read1_only=False`` (both mates of a BAM pair as fragments) is not served: the decoders keep
    read1 records only, which is what every caller on the hot path asks for."""

    def __init__(self, path, reference_file=None, threads: int = 1, quality_threshold: int = 30,
                 read1_only: bool = True) -> None:
        from .exceptions import UnsupportedFormatError
        from .source import _check_path, open_source
        self.path = str(path) if isinstance(path, (str, Path)) else None
        self.reference_file = str(reference_file) if reference_file else None
        self.threads = threads
        self.quality_threshold = quality_threshold
        self.read1_only = read1_only
        self._src = None
        checked, is_bam = _check_path(path)  # the reference's open-time errors, in its order
        if is_bam and not read1_only:
            raise UnsupportedFormatError("read1_only=False is not supported: the BAM decoders emit one fragment per pair (read1)")
        self._is_sam, self._is_tabix = is_bam, not is_bam
        self._src = open_source(checked, threads)  # (warns when the rows are BED6, like the reference's format probe)
        self._bed_format = bool(self._src.bed6)
        self._chroms = {c: (self._src.lengths.get(c) if is_bam else None) for c in self._src.contigs}

    @property
    def chroms(self) -> Dict[str, Optional[int]]:
        """Contig name -> length (``None`` for tabix files, which carry no lengths)."""
        return self._chroms

    @property
    def is_sam(self) -> bool:
        return self._is_sam

    def fetch(self, contig: Optional[str] = None, start: Optional[int] = None,
              stop: Optional[int] = None) -> Generator[Fragment, None, None]:
        """``Fragment`` records of the region in file order, mapq-filtered (io/alignment.py:216-302).  ``contig=None``
        walks the whole file and ignores the bounds, as pysam does without a reference name."""
        if self._src is None:
            raise ValueError("I/O operation on closed file")
        from .source import get_engine
        src, eng = self._src, get_engine()
        src.check_fetch(contig, start, stop)
        if contig is None:
            src.load_all()
            todo = [(c, src.require(c), None, None) for c in src.contigs if c in src.loaded]
        else:
            todo = [(contig, src.require_interval(contig, start, stop, 1), start, stop)]
        for name, key, a, b in todo:
            s, e, q, st = eng.frag_select(key, a, b, self.quality_threshold, None, None, "fetch")
            for i in range(len(s)):
                yield Fragment(name, int(s[i]), int(e[i]), int(q[i]), bool(st[i]))

    def close(self) -> None:
        """Forget the source (device-resident tables stay cached for the next reader of the same file)."""
        self._src = None

    def __enter__(self) -> "AlignmentWrapper":
        return self

    def __exit__(self, exc_type, exc_val, exc_tb) -> None:
        self.close()
