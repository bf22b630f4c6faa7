"""
TEST INFRASTRUCTURE ONLY -- never imported by the product path.

Stand-in modules that let the *reference* package (``/root/reference/src``)
be imported in the build container, where its compiled third-party
dependencies (pysam, numba, pyBigWig, py2bit, loess) are absent.  Used only by
``oracle/gen_golden*.py`` to emit the golden vectors that ``tests/test_oracle_golden.py`` replays
against the restatement.  Nothing
here travels to the GPU box as a dependency of any test: the reference source
is not in this repository and ``/root/reference`` does not exist there.

The tabix stand-in implements the documented htslib region semantics the
reference relies on (``src/finaletoolkit/io/alignment.py:270-302``):

* rows are returned in file order as tuples of *strings* (``asTuple``),
* a region query ``(reference, start, end)`` returns rows of that contig with
  ``row.start < end and row.end > start`` (bounds ``None`` = open),
* ``reference=None`` iterates the whole file and ignores ``start``/``end``
  (pysam builds no region string without a reference),
* lines starting with the meta character ``#`` are skipped.
"""
from __future__ import annotations

import gzip
import sys
import types

REFERENCE_SRC = "/root/reference/src"


class _TabixFile:
    def __init__(self, filename, *args, **kwargs):
        self.filename = str(filename)
        self._rows = []
        with gzip.open(self.filename, "rt") as fh:
            for line in fh:
                if not line.strip() or line.startswith("#"):
                    continue
                self._rows.append(tuple(line.rstrip("\n").split("\t")))
        seen = []
        for row in self._rows:
            if row[0] not in seen:
                seen.append(row[0])
        self.contigs = seen
        # per-contig index for region queries on start-sorted files (the whole-genome DELFI goldens ask tens of
        # thousands of windows): rows of the contig, their starts, the longest row -- the filter below is unchanged,
        # the index only narrows the rows it is applied to
        self._by_contig = {}
        for row in self._rows:
            self._by_contig.setdefault(row[0], []).append(row)
        self._index = {}
        for name, rows in self._by_contig.items():
            try:
                st = [int(r[1]) for r in rows]
                ln = max(int(r[2]) - int(r[1]) for r in rows)
            except (ValueError, IndexError):
                continue
            if all(a <= b for a, b in zip(st, st[1:])):
                self._index[name] = (st, ln)

    def _candidates(self, reference, start, end):
        if reference is None:
            return self._rows
        rows = self._by_contig.get(reference, [])
        idx = self._index.get(reference)
        if idx is None:
            return rows
        import bisect
        st, ln = idx
        lo = 0 if start is None else bisect.bisect_left(st, start - ln)
        hi = len(rows) if end is None else bisect.bisect_left(st, end)
        return rows[lo:hi]

    def fetch(self, reference=None, start=None, end=None, region=None,
              parser=None, multiple_iterators=False):
        for row in self._candidates(reference, start, end):
            if reference is not None:
                if row[0] != reference:
                    continue
                try:
                    r_start = int(row[1])
                    r_end = int(row[2])
                except (ValueError, IndexError):
                    # htslib would have failed to index such a row; hand it on
                    # so the caller's own skip logic is exercised.
                    yield row
                    continue
                if end is not None and not (r_start < end):
                    continue
                if start is not None and not (r_end > start):
                    continue
            yield row

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class _FastaFile:
    """pysam.FastaFile stand-in: whole file in memory, ``fetch`` = plain slicing of the contig
    (0-based half-open), case preserved -- what htslib's faidx returns."""

    def __init__(self, filename, *args, **kwargs):
        self.filename = str(filename)
        seqs, name, parts = {}, None, []
        with open(self.filename) as fh:
            for line in fh:
                if line.startswith(">"):
                    if name is not None:
                        seqs[name] = "".join(parts)
                    name, parts = line[1:].split()[0], []
                else:
                    parts.append(line.strip())
        if name is not None:
            seqs[name] = "".join(parts)
        self._seqs = seqs
        self.references = tuple(seqs)
        self.lengths = tuple(len(v) for v in seqs.values())

    def fetch(self, reference=None, start=None, end=None, region=None):
        return self._seqs[reference][start:end]

    def close(self):
        pass


class _Dummy:
    def __init__(self, *a, **k):
        raise RuntimeError("not available in the oracle stub")


def install():
    """Insert the stand-in modules and put the reference on ``sys.path``."""
    if "pysam" not in sys.modules:
        pysam = types.ModuleType("pysam")
        pysam.TabixFile = _TabixFile
        try:
            from . import bamstub
        except ImportError:
            import bamstub
        pysam.AlignmentFile = bamstub.AlignmentFile
        pysam.AlignedSegment = bamstub.AlignedSegment
        pysam.FastaFile = _FastaFile
        pysam.asTuple = lambda *a, **k: None
        pysam.faidx = lambda *a, **k: None
        pysam.tabix_index = lambda *a, **k: None
        sys.modules["pysam"] = pysam

    if "numba" not in sys.modules:
        numba = types.ModuleType("numba")

        def jit(*jargs, **jkwargs):
            if len(jargs) == 1 and callable(jargs[0]) and not jkwargs:
                return jargs[0]
            return lambda fn: fn

        numba.jit = jit
        numba.njit = jit
        sys.modules["numba"] = numba

    for name in ("pyBigWig", "py2bit", "loess"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if "loess.loess_1d" not in sys.modules:
        sub = types.ModuleType("loess.loess_1d")
        sub.loess_1d = None
        sys.modules["loess.loess_1d"] = sub
        sys.modules["loess"].loess_1d = sub

    if REFERENCE_SRC not in sys.path:
        sys.path.insert(0, REFERENCE_SRC)
