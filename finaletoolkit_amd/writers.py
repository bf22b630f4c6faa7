"""
Output writers of the per-base features on the library's host threads (``csrc/ftk_writers.cpp``).

The reference prints one Python f-string per base (``frag/_wps.py:208-229``, ``frag/_multi_wps.py:328-341``) and
hands bigWig entries to pyBigWig (``:300-325``); after a 0.2 ms kernel that is seconds to minutes of
interpreter time for a chromosome.  Here the same bytes are formatted in C on all usable cores, ``.gz``
outputs are written as gzip members compressed in parallel (the decompressed stream is identical; a gzip
file carries a timestamp, so compressed bytes never were reproducible), and bigWig data sections are built
and deflated natively.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L

GZIP_LEVEL = 6


class TextBuffer:
    """Bytes formatted by the library (owned by it until ``free``)."""

    def __init__(self, lib, ptr, n):
        self._lib, self.ptr, self.n = lib, ptr, int(n)

    def tobytes(self) -> bytes:
        return C.string_at(self.ptr, self.n)

    def write(self, path: str, gzip_level: int = 0, append: bool = False, threads: int = 0):
        rc = self._lib.ftk_file_write(str(path).encode(), self.ptr, self.n, int(gzip_level), int(threads), int(append))
        if rc != L.FTK_OK:
            raise OSError(self._lib.ftk_fragtable_error().decode())

    def gzip_bytes(self, gzip_level: int = GZIP_LEVEL, threads: int = 0) -> bytes:
        """The gzip members ``write(path, gzip_level)`` would put into the file, as bytes (a rank that does not own
        the output file hands them to the one that does)."""
        if self.n == 0:
            return b""
        out, n = C.c_void_p(), C.c_int64()
        rc = self._lib.ftk_gzip_members(self.ptr, self.n, int(gzip_level), int(threads), C.byref(out), C.byref(n))
        if rc != L.FTK_OK:
            raise L.FtkError(rc, self._lib.ftk_fragtable_error().decode())
        try:
            return C.string_at(out.value, n.value)
        finally:
            self._lib.ftk_buffer_free(out.value)

    def free(self):
        if self.ptr:
            self._lib.ftk_buffer_free(self.ptr)
            self.ptr = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.free()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _call(fn, *args) -> TextBuffer:
    lib = L.load()
    out, n = C.c_void_p(), C.c_int64()
    rc = getattr(lib, fn)(*args, C.byref(out), C.byref(n))
    if rc != L.FTK_OK:
        raise L.FtkError(rc, lib.ftk_fragtable_error().decode())
    return TextBuffer(lib, out.value, n.value)


def wig_body(values, threads: int = 0) -> TextBuffer:
    """``"".join(f"{v}\\n" for v in values)`` for int64 values."""
    v = np.ascontiguousarray(values, dtype=np.int64)
    return _call("ftk_format_wig_i64", L.ptr(v), len(v), int(threads))


def _runs(starts, offsets, n_values):
    st = np.ascontiguousarray(starts, dtype=np.int64)
    if offsets is None:
        offs = np.array([0, n_values], np.int64)
    else:
        offs = np.ascontiguousarray(offsets, dtype=np.int64)
    if len(offs) != len(st) + 1:
        raise ValueError("offsets must hold one entry more than starts")
    return st, offs


def bedgraph_rows(contig: str, starts, values, offsets=None, threads: int = 0) -> TextBuffer:
    """``contig  pos  pos+1  value`` per base of every run (run ``k`` = ``values[offsets[k]:offsets[k+1]]``
    starting at ``starts[k]``; one run when ``offsets`` is None and ``starts`` a single position); int64 values
    as ``str(int)``, float64 values as ``repr(float)``."""
    v = np.asarray(values)
    flt = v.dtype.kind == "f"
    v = np.ascontiguousarray(v, dtype=np.float64 if flt else np.int64)
    st, offs = _runs(np.atleast_1d(starts), offsets, len(v))
    return _call("ftk_format_bedgraph_f64" if flt else "ftk_format_bedgraph_i64", str(contig).encode(), L.ptr(st),
                 L.ptr(offs), len(st), L.ptr(v), int(threads))


def bedgraph_batches(contig: str, starts, values, offsets=None, max_values: int = 8 << 20, threads: int = 0):
    """``bedgraph_rows`` in pieces of at most ``max_values`` rows (about 25 bytes each), cut at run boundaries
    and inside runs longer than that: a generator of ``TextBuffer`` whose concatenation is the full text, so a
    whole chromosome of per-base rows never sits in memory at once."""
    v = np.asarray(values)
    st, offs = _runs(np.atleast_1d(starts), offsets, len(v))
    k, n_runs = 0, len(st)
    while k < n_runs:
        a = int(offs[k])
        if offs[k + 1] - a > max_values:  # one long run: its own pieces
            for o in range(a, int(offs[k + 1]), max_values):
                e = min(o + max_values, int(offs[k + 1]))
                yield bedgraph_rows(contig, [int(st[k]) + (o - a)], v[o:e], None, threads)
            k += 1
            continue
        j = int(np.searchsorted(offs, a + max_values, side="right")) - 1  # last run end within the budget
        j = min(max(j, k + 1), n_runs)
        yield bedgraph_rows(contig, st[k:j], v[a:int(offs[j])], offs[k:j + 1] - a, threads)
        k = j


def write_text(path: str, data: bytes, gzip_level: int = 0, append: bool = False, threads: int = 0):
    """Plain or gzip-member write of a Python bytes object (headers and other small pieces)."""
    lib = L.load()
    rc = lib.ftk_file_write(str(path).encode(), data, len(data), int(gzip_level), int(threads), int(append))
    if rc != L.FTK_OK:
        raise OSError(lib.ftk_fragtable_error().decode())


def bigwig_sections(chrom_id: int, starts, values, offsets=None, items_per_section: int = 16384, level: int = 6,
                    threads: int = 0):
    """fixedStep data sections of one ``addEntries(chrom, start, values=..., span=1, step=1)`` call per run.
    Returns ``(blob bytes, table int64[n_sec, 3] = start end compressed_bytes, stats float64[n_sec, 4] = min max
    sum sumsq)``."""
    lib = L.load()
    v = np.asarray(values)
    kind = 1 if v.dtype.kind == "f" else 0
    v = np.ascontiguousarray(v, dtype=np.float64 if kind else np.int64)
    st, offs = _runs(np.atleast_1d(starts), offsets, len(v))
    out, out_len, n_sec = C.c_void_p(), C.c_int64(), C.c_int64()
    table, stats = C.c_void_p(), C.c_void_p()
    rc = lib.ftk_bigwig_fixedstep_sections(int(chrom_id), L.ptr(st), L.ptr(offs), len(st), L.ptr(v), kind,
                                           int(items_per_section), int(level), int(threads), C.byref(out),
                                           C.byref(out_len), C.byref(n_sec), C.byref(table), C.byref(stats))
    if rc != L.FTK_OK:
        raise L.FtkError(rc, lib.ftk_fragtable_error().decode())
    try:
        blob = C.string_at(out.value, out_len.value)
        k = int(n_sec.value)
        tab = np.ctypeslib.as_array(C.cast(table, C.POINTER(C.c_int64)), (max(k, 1) * 3,))[:k * 3].reshape(k, 3).copy()
        sts = np.ctypeslib.as_array(C.cast(stats, C.POINTER(C.c_double)), (max(k, 1) * 4,))[:k * 4].reshape(k, 4).copy()
    finally:
        for q in (out, table, stats):
            lib.ftk_buffer_free(q.value)
    return blob, tab, sts


def frag_rows(contig: str, start, end, mapq, strand, bed6: bool = False, threads: int = 0) -> TextBuffer:
    """Fragment-file rows ``contig  start  end  mapq  +|-`` (``bed6``: a ``.`` name column before mapq)."""
    s_ = np.ascontiguousarray(start, dtype=np.int32)
    e_ = np.ascontiguousarray(end, dtype=np.int32)
    q_ = np.ascontiguousarray(mapq, dtype=np.uint8)
    t_ = np.ascontiguousarray(strand, dtype=np.uint8)
    if not (len(s_) == len(e_) == len(q_) == len(t_)):
        raise ValueError("fragment columns differ in length")
    return _call("ftk_format_frag_rows", str(contig).encode(), L.ptr(s_), L.ptr(e_), L.ptr(q_), L.ptr(t_), len(s_),
                 int(bool(bed6)), int(threads))


def bgzf_write(path: str, data, level: int = 6, append: bool = False, write_eof: bool = True, threads: int = 0):
    """Write ``data`` (bytes or a ``TextBuffer``) as BGZF blocks compressed in parallel; returns the file offset
    of every data block plus the offset behind the last one (int64 array)."""
    lib = L.load()
    if isinstance(data, TextBuffer):
        ptr, n = C.c_void_p(data.ptr), data.n
    else:
        ptr, n = data, len(data)
    offs = np.zeros(-(-n // 0xFF00) + 1, np.int64)
    rc = lib.ftk_bgzf_write(str(path).encode(), ptr, n, int(level), int(threads), int(append), int(write_eof), L.ptr(offs))
    if rc != L.FTK_OK:
        raise OSError(lib.ftk_fragtable_error().decode())
    return offs


# ---------------------------------------------------------------------------------------------------------------
# Row files of the interval commands.  What the formats ARE is the reference's contract (suffix rules, column order,
# header lines, `str()` / `repr()` of every value: frag/_coverage.py:262-296, frag/_frag_length.py:466-490,607-632);
# how they are produced is ours: the rows of a call are one string, written once (`.gz`: as gzip members, compressed
# by the library's threads - the decompressed stream is what `gzip.open(..., "wt")` would have held).
# ---------------------------------------------------------------------------------------------------------------
def _emit(output_file: str, text: str, plain: tuple, gz: tuple, bad_suffix: str) -> None:
    """``text`` to ``output_file``: ``"-"`` = stdout, a name ending in one of ``gz`` = gzip, in one of ``plain`` = as
    is (``plain=None``: any other name), else ``ValueError(bad_suffix)``."""
    import sys
    if output_file == "-":
        sys.stdout.write(text)
    elif gz and output_file.endswith(gz):
        write_text(output_file, text.encode(), GZIP_LEVEL)
    elif plain is None or output_file.endswith(plain):
        with open(output_file, "w") as fh:
            fh.write(text)
    else:
        raise ValueError(bad_suffix)


def check_suffix(output_file: str, suffixes: tuple, message: str) -> None:
    if not (output_file == "-" or output_file.endswith(suffixes)):
        raise ValueError(message)


def write_coverage_rows(output_file: str, intervals, values) -> None:
    """``contig start stop name value`` per interval (``.bedgraph``: no name column); ``.bed`` / ``.bedgraph`` /
    ``.bed.gz`` / ``-``."""
    message = "output_file should have .bed or .bed.gz as suffix"
    check_suffix(output_file, (".bed", ".bedgraph", ".bed.gz"), message)
    if output_file.endswith(".bedgraph"):
        text = "".join([f"{c}\t{a}\t{b}\t{v}\n" for (c, a, b, _), v in zip(intervals, values)])
    else:
        text = "".join([f"{c}\t{a}\t{b}\t{n}\t{v}\n" for (c, a, b, n), v in zip(intervals, values)])
    _emit(output_file, text, (".bed", ".bedgraph"), (".bed.gz",), message)


def write_length_bins(output_file: str, bins, counts, bin_size: int, stats=None) -> None:
    """The ``min max count`` table of ``frag_length_bins`` (+ ``#name: value`` lines when ``stats`` is given); any
    file name, ``.gz`` = gzip, ``-`` = stdout."""
    rows = ["min\tmax\tcount\n"]
    rows += [f"{lo}\t{lo + bin_size - 1}\t{n}\n" for lo, n in zip(bins, counts)]
    if stats:
        rows += [f"#{name}: {value}\n" for name, value in stats]
    _emit(output_file, "".join(rows), None, (".gz",), "")


def write_length_stats(output_file: str, results, short_reads: int, lines=None) -> None:
    """The per-interval statistics table of ``frag_length_intervals``: a header line, then the eleven fields of every
    result, tab-separated, ``str()`` of each (``lines``: the rows already formatted, one per result); ``.bed`` /
    ``.bedgraph`` / ``.bed.gz`` / ``-``."""
    message = "The output file should have .bed or .bed.gz as as suffix."
    check_suffix(output_file, (".bed", ".bedgraph", ".bed.gz"), message)
    head = f"contig\tstart\tstop\tname\tmean\tmedian\tstdev\tmin\tmax\tcount\ts{short_reads}\n"
    if lines is None or any(ln is None for ln in lines):
        lines = [f"{r[0]}\t{r[1]}\t{r[2]}\t{r[3]}\t{r[4]}\t{r[5]}\t{r[6]}\t{r[7]}\t{r[8]}\t{r[9]}\t{r[10]}" for r in results]
    body = "\n".join(lines)
    _emit(output_file, head + body + "\n", (".bed", ".bedgraph"), (".bed.gz",), message)
