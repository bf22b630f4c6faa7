"""
Fragment sources: open a BAM / tabix-indexed fragment file once, decode it with
the C++ decoders (``csrc/ftk_decode.cpp``) and keep every contig's SoA resident
in HBM on the process-wide :class:`Engine`.

This is the counterpart of the reference's ``AlignmentWrapper``
(``src/finaletoolkit/io/alignment.py:74-302``): same accepted inputs by
extension, same index-presence checks and exception types, same BED6 warning --
but instead of handing out a per-window Python iterator it hands out contigs
that are already on the GPU.
"""
from __future__ import annotations

import ctypes as C
import os
import warnings
from collections import OrderedDict
from pathlib import Path
from typing import Optional

import numpy as np

from . import _lib as L
from .engine import Engine
from .exceptions import MissingIndexError, UnsupportedFormatError

_ENGINE: Optional[Engine] = None
_SOURCES: "OrderedDict[tuple, FragSource]" = OrderedDict()
_MAX_SOURCES = 4
_NEXT_ID = 0
REGION_READS: list = []  # (path, contig, start, stop) of every region stream opened by this process (diagnostics, tests)


def get_engine() -> Engine:
    """Process-wide engine on ``FTK_DEVICE`` (else ``LOCAL_RANK``, else 0)."""
    global _ENGINE
    if _ENGINE is None:
        dev = int(os.environ.get("FTK_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        _ENGINE = Engine(dev)
    return _ENGINE


_SIDE_ENGINE: Optional[Engine] = None


def get_side_engine() -> Engine:
    """A SECOND context on the engine's device, for work that runs beside the main one from another thread (a ctx is
    single-threaded; different ctxs may be driven from different threads - include/ftk.h).  ``frag.delfi`` counts the
    bins' G + C on it while the fragment file is still being decoded.  Kept for the process (its reference blocks and
    page-locked staging are recycled like the main engine's), dropped by ``close_all``."""
    global _SIDE_ENGINE
    if _SIDE_ENGINE is None:
        _SIDE_ENGINE = Engine(get_engine().device)
    return _SIDE_ENGINE


def usable_cores() -> int:
    """Host cores this process may really use: scheduler affinity, capped by the cgroup CPU quota (a
    container can see 256 logical CPUs and own 16), divided by ``LOCAL_WORLD_SIZE`` when this process is one rank
    of several on the node; ``FTK_HOST_THREADS`` overrides."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period) + 0.5)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(quota / period + 0.5)))
        except (OSError, ValueError):
            pass
    # one rank per GPU on one node: a rank's share of the host cores (the native side applies the same rule,
    # ftk_host::default_threads)
    try:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", "1"))
    except ValueError:
        local_world = 1
    if local_world > 1:
        n = max(1, n // local_world)
    if os.environ.get("FTK_HOST_THREADS", "").isdigit() and int(os.environ["FTK_HOST_THREADS"]) > 0:
        n = int(os.environ["FTK_HOST_THREADS"])
    return max(1, n)


def decode_threads(workers: int | None = None) -> int:
    """Decoder threads: ``workers`` when given, else the usable cores (at most 64)."""
    if workers and workers > 0:
        return int(workers)
    return max(1, min(64, usable_cores()))



def _check_skipped(lib, stream, path: str, seen: list) -> None:
    """After an ``ftk_fragstream_next`` on a BAM stream: act on the records the decoder met that the reference does
    not simply skip (``include/ftk.h``: ``ftk_fragstream_skipped``).  A CIGAR-less read1 with TLEN < 0 makes the
    reference raise ``TypeError`` (``None + tlen``, io/alignment.py:257) - so does this; a fragment with a negative
    start is yielded by the reference and cannot be held by the int32 columns here - dropped, with a warning."""
    out = (C.c_int64 * 2)()
    if lib.ftk_fragstream_skipped(stream, C.byref(out)) != L.FTK_OK:
        return
    if out[1] > 0:
        raise TypeError("unsupported operand type(s) for +: 'NoneType' and 'int' "
                        f"({path}: {out[1]} read1 record(s) without a CIGAR and with TLEN < 0 - pysam's reference_end "
                        "is None for them, and the reference fails the same way at io/alignment.py:257)")
    if out[0] > seen[0]:
        warnings.warn(f"{path}: {out[0] - seen[0]} read1 record(s) whose fragment starts before position 0 "
                      "(reference_end + TLEN < 0) or lies beyond 2^31 were dropped; the reference keeps such "
                      "fragments", UserWarning, stacklevel=3)
        seen[0] = out[0]

class FragSource:
    """One input file; contig ``c`` lives on the engine as ``key(c)`` once it has been decoded.

    Files with a usable index (tabix ``.tbi`` / ``.bai``) are opened lazily: the contig list comes from the
    index or the BAM header and a contig is decoded -- seeking through the index, so only its own blocks are
    read -- the first time a feature asks for it; whole-file operations call ``load_all``.  Files without a
    usable index are decoded in one streaming pass when they are opened."""

    def __init__(self, path: str, is_bam: bool, bed6: bool, contigs, lengths, uid: int, lazy: bool = False,
                 workers: int | None = None):
        self.path = path
        self.is_bam = is_bam
        self.bed6 = bed6
        self.contigs = list(contigs)              # file / header order
        self.lengths = dict(lengths)              # name -> length (BAM) / None
        self.uid = uid
        self.loaded = set()
        self.regions = set()                      # engine keys of partial tables (require_region)
        self._interval_hits = {}                  # contig -> single-interval calls that found it not resident
        self._recent_regions = []                 # region tables kept for require_interval (most recent last)
        self.lazy = lazy
        self.workers = workers
        self.decode_stage_ms = None               # set by stream_source when the whole file has been decoded

    def key(self, contig: str) -> str:
        return f"{self.uid}:{contig}"

    @property
    def chroms(self):
        return {c: self.lengths.get(c) for c in self.contigs}

    def has(self, contig: str) -> bool:
        return contig in self.loaded or (self.lazy and contig in self.contigs)

    def check_fetch(self, contig, start, stop) -> None:
        """pysam's argument checks for ``AlignmentFile.fetch(contig, start, stop)`` (libcalignmentfile.pyx
        ``parse_region``), which the reference's BAM path inherits (io/alignment.py:245): a negative start and
        ``start > stop`` are ``ValueError``s.  (Unknown contigs: ``require``.)  Tabix input is not checked here - the
        reference's tabix path is pinned with bounds it accepts only."""
        if not self.is_bam or contig is None:
            return
        a = 0 if start is None else int(start)
        b = (1 << 31) - 1 if stop is None else int(stop)
        if a > b:
            raise ValueError(f"invalid coordinates: start ({a}) > stop ({b})")
        if not 0 <= a < (1 << 31) - 1:
            raise ValueError(f"start out of range ({a})")
        if not 0 <= b <= (1 << 31) - 1:
            raise ValueError(f"stop out of range ({b})")

    def require(self, contig: str) -> str:
        """Engine key of ``contig`` (decoding it now if the source is lazy); ValueError if the file has no
        such contig (pysam raises ValueError for an unknown region, which the reference lets propagate)."""
        if contig not in self.loaded:
            if not (self.lazy and contig in self.contigs):
                raise ValueError(f"could not create iterator for region '{contig}': contig not present in {self.path}")
            self._load_one(contig)
        return self.key(contig)

    def _load_one(self, contig: str):
        eng = get_engine()
        lib = L.load()
        stream = C.c_void_p()
        # text rows are parsed on the engine's GPU (ftk_fragstream_open_device; BAM records on the host)
        rc = lib.ftk_fragstream_open_device(eng.device, self.path.encode(), contig.encode(), int(self.is_bam),
                                            decode_threads(self.workers), 1, C.byref(stream))
        if rc != L.FTK_OK:
            raise UnsupportedFormatError(lib.ftk_fragtable_error().decode())
        try:
            table = C.c_void_p()
            rc = lib.ftk_fragstream_next(stream, C.byref(table))
            if rc != L.FTK_OK:
                raise UnsupportedFormatError(lib.ftk_fragtable_error().decode())
            if self.is_bam:
                try:
                    _check_skipped(lib, stream, self.path, [0])
                except TypeError:
                    if table.value:
                        lib.ftk_fragtable_free(table)
                    raise
            if table.value:
                try:
                    eng.load_contig_from_table(self.key(contig), table, 0, self.is_bam)
                finally:
                    lib.ftk_fragtable_free(table)
            else:  # listed, but without a usable row / read: valid and empty
                e32, e8 = np.zeros(0, np.int32), np.zeros(0, np.uint8)
                eng.load_contig(self.key(contig), e32, e32, e8, e8, *((e32, e32) if self.is_bam else ()))
            self.loaded.add(contig)
        finally:
            lib.ftk_fragstream_close(stream)

    def require_region(self, contig: str, start: int, stop: int) -> str:
        """Engine key of a table that holds EVERY fragment of ``contig`` overlapping ``[start, stop)`` - the whole
        contig when that is resident already (or when the file cannot be entered in the middle of a contig: an index
        without the 16 kb linear index), else the rows / records ``ftk_fragstream_open_region`` reads for the region
        (index seek to its first row, the parsed rows say where it is complete; for a BAM: every fragment whose read1
        overlaps the region).  A rank of a multi-GPU run loads its share of a
        contig this way (``sharding.split_counts``); ``release_region`` drops the table."""
        if contig in self.loaded or not self.lazy:
            return self.require(contig)
        if contig not in self.contigs:
            raise ValueError(f"could not create iterator for region '{contig}': contig not present in {self.path}")
        start, stop = max(0, int(start)), int(stop)
        if stop <= start:
            return self.require(contig)
        key = f"{self.uid}:{contig}@{start}-{stop}"
        if key in self.regions:
            return key
        eng = get_engine()
        lib = L.load()
        stream = C.c_void_p()
        REGION_READS.append((self.path, contig, start, stop))
        if len(REGION_READS) > 4096:  # (a diagnostic trail, not a log: a long-lived host keeps the most recent ones)
            del REGION_READS[:2048]
        rc = lib.ftk_fragstream_open_region(eng.device, self.path.encode(), contig.encode(), start, stop, int(self.is_bam),
                                            decode_threads(self.workers), 1, C.byref(stream))
        if rc != L.FTK_OK:
            raise UnsupportedFormatError(lib.ftk_fragtable_error().decode())
        try:
            table = C.c_void_p()
            rc = lib.ftk_fragstream_next(stream, C.byref(table))
            if rc != L.FTK_OK:
                raise UnsupportedFormatError(lib.ftk_fragtable_error().decode())
            if self.is_bam:
                try:
                    _check_skipped(lib, stream, self.path, [0])
                except TypeError:
                    if table.value:
                        lib.ftk_fragtable_free(table)
                    raise
            if table.value:
                try:
                    eng.load_contig_from_table(key, table, 0, self.is_bam)
                finally:
                    lib.ftk_fragtable_free(table)
            else:
                e32, e8 = np.zeros(0, np.int32), np.zeros(0, np.uint8)
                eng.load_contig(key, e32, e32, e8, e8, *((e32, e32) if self.is_bam else ()))
            self.regions.add(key)
        finally:
            lib.ftk_fragstream_close(stream)
        return key

    def require_interval(self, contig: str, start, stop, pad: int = 0) -> str:
        """Engine key for ONE call about ``contig:[start - pad, stop + pad)`` (``frag.wps``, ``single_coverage``,
        ``frag_length`` ... on an interval).  A contig that is resident is used as it is; one that is not is not
        decoded whole for a first or second small interval (a 30x chr1 is 25 M rows for a call that needs a few
        thousand) - the interval's rows come through the index as a region, the two most recent region tables are
        kept - but from the third call on it is, since whoever asks three times will ask again."""
        if start is None or stop is None or contig in self.loaded or not self.lazy:
            return self.require(contig)
        start, stop = int(start) - int(pad), int(stop) + int(pad)
        hits = self._interval_hits.get(contig, 0) + 1
        self._interval_hits[contig] = hits
        if hits >= 3 or stop - start > 20_000_000:
            return self.require(contig)
        key = self.require_region(contig, start, stop)
        if key in self.regions:
            if key in self._recent_regions:
                self._recent_regions.remove(key)
            self._recent_regions.append(key)
            while len(self._recent_regions) > 2:
                self.release_region(self._recent_regions.pop(0))
        return key

    def release_region(self, key: str):
        if key in self.regions:
            self.regions.discard(key)
            eng = get_engine()
            if eng.has_contig(key):
                eng.release(key)

    def load_all(self):
        """Make every contig resident (whole-file operations).  When most of the file is still missing, ONE
        streaming pass over it (decode of contig k+1 overlapped with the upload of contig k) replaces one index
        seek and stream per contig."""
        if not self.lazy:
            return
        missing = [c for c in self.contigs if c not in self.loaded]
        if len(missing) > 1 and 2 * len(missing) >= len(self.contigs):
            self._stream_missing(set(missing))
        for c in self.contigs:
            if c not in self.loaded:
                self._load_one(c)

    def _stream_missing(self, missing):
        for _ in self._stream_missing_iter(missing):
            pass

    def _stream_missing_iter(self, missing):
        """One streaming pass over the file; yields the name of every contig of ``missing`` as soon as it is
        resident (the decoder is already in the next contig)."""
        eng = get_engine()
        lib = L.load()
        stream = C.c_void_p()
        rc = lib.ftk_fragstream_open_device(eng.device, self.path.encode(), None, int(self.is_bam),
                                            decode_threads(self.workers), 2, C.byref(stream))
        if rc != L.FTK_OK:
            raise UnsupportedFormatError(lib.ftk_fragtable_error().decode())
        seen = [0]
        try:
            while True:
                table = C.c_void_p()
                rc = lib.ftk_fragstream_next(stream, C.byref(table))
                if rc != L.FTK_OK:
                    raise UnsupportedFormatError(lib.ftk_fragtable_error().decode())
                if self.is_bam:
                    try:
                        _check_skipped(lib, stream, self.path, seen)
                    except TypeError:
                        if table.value:
                            lib.ftk_fragtable_free(table)
                        raise
                if not table.value:
                    break
                try:
                    name = lib.ftk_fragtable_contig_name(table, 0).decode()
                    fresh = name in missing and name not in self.loaded
                    if fresh:
                        eng.load_contig_from_table(self.key(name), table, 0, self.is_bam)
                        self.loaded.add(name)
                finally:
                    lib.ftk_fragtable_free(table)
                if fresh:
                    yield name
        finally:
            lib.ftk_fragstream_close(stream)

    def release(self):
        eng = get_engine()
        for c in list(self.loaded):
            if eng.has_contig(self.key(c)):
                eng.release(self.key(c))
        self.loaded.clear()
        for key in list(self.regions):
            if eng.has_contig(key):
                eng.release(key)
        self.regions.clear()


def _check_path(input_file) -> tuple[str, bool]:
    """Mirror AlignmentWrapper.__init__/_open_file (io/alignment.py:103-203)."""
    if not isinstance(input_file, (str, Path)):
        raise UnsupportedFormatError(
            "finaletoolkit_amd reads BAM and tabix-indexed fragment files by path; "
            f"open pysam handles are not supported (got {type(input_file).__name__})")
    path = str(input_file)
    if not os.path.exists(path):
        raise FileNotFoundError(f"Alignment file not found: {input_file}")
    lower = path.lower()
    if lower.endswith((".bam", ".cram", ".sam")):
        if lower.endswith(".bam"):
            if not (os.path.exists(path + ".bai") or os.path.exists(path[:-4] + ".bai")):
                raise MissingIndexError(f"BAM file {path} missing index (.bai)")
            return path, True
        if lower.endswith(".cram"):
            if not (os.path.exists(path + ".crai") or os.path.exists(path[:-5] + ".crai")):
                raise MissingIndexError(f"CRAM file {path} missing index (.crai)")
        raise UnsupportedFormatError(
            f"{path}: CRAM/SAM decoding needs htslib codecs and is not implemented in finaletoolkit_amd; "
            "convert to BAM or a fragment file")
    if lower.endswith((".gz", ".bgz")):
        if not os.path.exists(path + ".tbi"):
            raise MissingIndexError(f"Compressed file {path} missing tabix index (.tbi)")
        return path, False
    raise UnsupportedFormatError(f"Unsupported file format: {path}")


def _columns(lib, table, i, rows):
    ps = [C.c_void_p() for _ in range(6)]
    rc = lib.ftk_fragtable_columns(table, i, *[C.byref(p) for p in ps])
    if rc != L.FTK_OK:
        raise L.FtkError(rc, lib.ftk_fragtable_error().decode())

    def arr(p, ctype):
        if not p.value or rows == 0:
            return None
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(ctype)), (rows,))

    return (arr(ps[0], C.c_int32), arr(ps[1], C.c_int32), arr(ps[2], C.c_uint8), arr(ps[3], C.c_uint8),
            arr(ps[4], C.c_int32), arr(ps[5], C.c_int32))


def stream_source(input_file, workers: int | None = None, queued: int = 2):
    """Decode ``input_file`` with the streaming decoder (``ftk_fragstream_*``) and upload it contig by
    contig: a generator yielding ``(src, contig)`` as soon as a contig is resident in HBM, while the
    decoder threads already work on the next one (decode || H2D || the caller's kernels).  Host memory is
    bounded by ``queued`` contigs whatever the file size.  The finished source is cached like
    ``open_source``'s; if the file is already cached its contigs are yielded straight away."""
    global _NEXT_ID
    path, is_bam = _check_path(input_file)
    st = os.stat(path)
    ckey = (os.path.abspath(path), st.st_mtime_ns, st.st_size)
    src = _SOURCES.get(ckey)
    if src is not None:
        _SOURCES.move_to_end(ckey)
        src.load_all()
        for c in src.contigs:
            if c in src.loaded:
                yield src, c
        return
    eng = get_engine()  # fails loudly without the HIP library / a GPU
    lib = L.load()
    stream = C.c_void_p()
    # text rows are parsed on the engine's GPU (ftk_fragstream_open_device; BAM records on the host)
    rc = lib.ftk_fragstream_open_device(eng.device, path.encode(), None, int(is_bam), decode_threads(workers), int(queued),
                                        C.byref(stream))
    if rc != L.FTK_OK:
        msg = lib.ftk_fragtable_error().decode()
        raise FileNotFoundError(msg) if rc == L.FTK_ERR_IO else UnsupportedFormatError(msg)
    src = FragSource(path, is_bam, False, [], {}, _NEXT_ID)
    _NEXT_ID += 1
    complete = False
    try:
        if is_bam:  # header order and lengths are known up front
            for i in range(lib.ftk_fragstream_n_refs(stream)):
                name = lib.ftk_fragstream_ref_name(stream, i).decode()
                src.contigs.append(name)
                src.lengths[name] = lib.ftk_fragstream_ref_length(stream, i)
        seen = [0]
        while True:
            table = C.c_void_p()
            rc = lib.ftk_fragstream_next(stream, C.byref(table))
            if rc != L.FTK_OK:
                msg = lib.ftk_fragtable_error().decode()
                raise FileNotFoundError(msg) if rc == L.FTK_ERR_IO else UnsupportedFormatError(msg)
            if is_bam:
                try:
                    _check_skipped(lib, stream, path, seen)
                except TypeError:
                    if table.value:
                        lib.ftk_fragtable_free(table)
                    raise
            if not table.value:
                break
            try:
                name = lib.ftk_fragtable_contig_name(table, 0).decode()
                src.bed6 = bool(lib.ftk_fragtable_is_bed6(table))
                if not is_bam:
                    src.contigs.append(name)
                    src.lengths[name] = None
                eng.load_contig_from_table(src.key(name), table, 0, is_bam)
                src.loaded.add(name)
            finally:
                lib.ftk_fragtable_free(table)
            yield src, name
        if is_bam:  # contigs of the header without a single usable read: valid, empty (pysam yields nothing)
            empty32, empty8 = np.zeros(0, np.int32), np.zeros(0, np.uint8)
            for name in src.contigs:
                if name not in src.loaded:
                    eng.load_contig(src.key(name), empty32, empty32, empty8, empty8, empty32, empty32)
                    src.loaded.add(name)
        stage = (C.c_double * 6)()
        if lib.ftk_fragstream_stage_ms(stream, C.byref(stage)) == L.FTK_OK:
            # wall time of the producer thread per stage (ms): what the decode of this file cost
            src.decode_stage_ms = dict(zip(("read", "inflate", "parse", "merge", "emit", "other"), (round(x, 2) for x in stage)))
        complete = True
    finally:
        lib.ftk_fragstream_close(stream)
        if not complete:  # consumer stopped early or the decode failed: do not cache a partial source
            src.release()
    _SOURCES[ckey] = src
    while len(_SOURCES) > _MAX_SOURCES:
        _, old = _SOURCES.popitem(last=False)
        old.release()


def _lazy_source(path: str, is_bam: bool, workers) -> Optional[FragSource]:
    """A lazily loaded source when the file's index is usable, else None."""
    global _NEXT_ID
    lib = L.load()
    if is_bam:
        index = path + ".bai" if os.path.exists(path + ".bai") else path[:-4] + ".bai"
        try:
            if os.path.getsize(index) < 8 or open(index, "rb").read(4) != b"BAI\x01":
                return None
        except OSError:
            return None
        stream = C.c_void_p()  # header only: a contig filter no header can contain
        rc = lib.ftk_fragstream_open(path.encode(), b"\x01", 1, 1, 1, C.byref(stream))
        if rc != L.FTK_OK:
            return None
        try:
            n = lib.ftk_fragstream_n_refs(stream)
            names = [lib.ftk_fragstream_ref_name(stream, i).decode() for i in range(n)]
            lengths = {names[i]: lib.ftk_fragstream_ref_length(stream, i) for i in range(n)}
            table = C.c_void_p()
            rc = lib.ftk_fragstream_next(stream, C.byref(table))
            if table.value:
                lib.ftk_fragtable_free(table)
        finally:
            lib.ftk_fragstream_close(stream)
        if rc != L.FTK_OK or not names:
            return None
        src = FragSource(path, True, False, names, lengths, _NEXT_ID, lazy=True, workers=workers)
    else:
        need, bed6 = C.c_int64(), C.c_int()
        if lib.ftk_fragfile_index_contigs(path.encode(), None, 0, C.byref(need), C.byref(bed6)) != L.FTK_OK:
            return None
        buf = C.create_string_buffer(max(int(need.value), 1))
        if lib.ftk_fragfile_index_contigs(path.encode(), buf, len(buf), C.byref(need), None) != L.FTK_OK:
            return None
        names = [n for n in buf.value.decode().split("\n") if n]
        src = FragSource(path, False, bool(bed6.value), names, {n: None for n in names}, _NEXT_ID, lazy=True,
                         workers=workers)
    _NEXT_ID += 1
    return src


def open_source(input_file, workers: int | None = None, warn_bed6: bool = True) -> FragSource:
    """The (cached) source of ``input_file``.  With a usable index nothing is decoded yet (``FragSource``);
    otherwise the file is decoded now, streaming contig by contig (``stream_source``), so the host never
    holds more than two contigs of it.  ``FTK_LAZY_SOURCE=0`` forces the one-pass decode."""
    path, is_bam = _check_path(input_file)
    st = os.stat(path)
    ckey = (os.path.abspath(path), st.st_mtime_ns, st.st_size)
    src = _SOURCES.get(ckey)
    if src is None:
        get_engine()  # fails loudly without the HIP library / a GPU
        if os.environ.get("FTK_LAZY_SOURCE", "1") != "0":
            src = _lazy_source(path, is_bam, workers)
        if src is not None:
            _SOURCES[ckey] = src
            while len(_SOURCES) > _MAX_SOURCES:
                _, old = _SOURCES.popitem(last=False)
                old.release()
        else:
            for _ in stream_source(input_file, workers):
                pass
            src = _SOURCES[ckey]
    else:
        _SOURCES.move_to_end(ckey)
    if src.bed6 and warn_bed6:
        _warn_bed6()
    return src


def resident_contigs(input_file, names, workers: int | None = None, stream_all: bool = True, warn_bed6: bool = True,
                     queued: int = 2):
    """Generator of ``(src, contig)`` over the contigs of ``names`` the file holds, each yielded as soon as it is
    resident in HBM -- the way in for whole-genome drivers (``frag.delfi``): the caller's kernels for contig k run
    while the decoder is already in contig k+1.  A file without a usable index is decoded in ONE streaming pass
    (``stream_source``); a lazily indexed one in one streaming pass too when ``stream_all`` and most of it is
    wanted, else contig by contig through the index (a rank of a multi-GPU run reads only its own blocks); a
    cached source yields straight away.  Contigs the file lacks are not yielded (the caller decides)."""
    path, is_bam = _check_path(input_file)
    st = os.stat(path)
    ckey = (os.path.abspath(path), st.st_mtime_ns, st.st_size)
    src = _SOURCES.get(ckey)
    everything = names is None  # (every contig of the file: a caller that starts the decode before it knows its plan)
    wanted = list(dict.fromkeys(names)) if not everything else None
    want_set = set(wanted) if not everything else None
    if src is None:
        get_engine()  # fails loudly without the HIP library / a GPU
        if os.environ.get("FTK_LAZY_SOURCE", "1") != "0":
            src = _lazy_source(path, is_bam, workers)
            if src is not None:
                _SOURCES[ckey] = src
                while len(_SOURCES) > _MAX_SOURCES:
                    _, old = _SOURCES.popitem(last=False)
                    old.release()
    else:
        _SOURCES.move_to_end(ckey)
    if src is not None:
        if src.bed6 and warn_bed6:
            _warn_bed6()
        have = [c for c in (src.contigs if everything else wanted) if src.has(c)]
        missing = [c for c in have if c not in src.loaded]
        done = set()
        if src.lazy and stream_all and len(missing) > 1 and 2 * len(missing) >= len(src.contigs):
            for c in src._stream_missing_iter(set(missing)):
                done.add(c)
                yield src, c
        for c in have:
            if c not in done:
                src.require(c)
                yield src, c
        return
    warned = False
    for src, c in stream_source(input_file, workers, queued):
        if src.bed6 and warn_bed6 and not warned:
            _warn_bed6()
            warned = True
        if everything or c in want_set:
            yield src, c


class EarlyContigs:
    """``resident_contigs(input_file, names, ...)`` started NOW, on a helper thread: the decoder opens the file and works
    towards the first contig while the caller is still reading its side files (``frag.delfi``: bins, blacklist, gap
    annotation, reference header - 17 ms of a 0.16 s whole-genome call).  Iterating joins the helper and carries on
    from the first contig; an error of the early part is raised there.  ``close()`` abandons it."""

    def __init__(self, input_file, workers=None, stream_all=True, warn_bed6=True, names=None, queued: int = 2):
        import threading
        # ``names``: the contigs the caller can want at all (its chrom.sizes) - a file with many more (decoys, alts) or
        # a caller that wants a few of them is then read through the index instead of being streamed whole
        self._gen = resident_contigs(input_file, names, workers, stream_all, warn_bed6, queued)
        self._first, self._err, self._end = None, None, False
        self._thread = threading.Thread(target=self._run, name="ftk-early-decode", daemon=True)
        self._thread.start()

    def _run(self):
        try:
            # (a BED6 warning of the first rows is raised from this thread: the warnings module's filters are
            # process-wide, so it reaches the caller's `catch_warnings` / `pytest.warns` all the same - and touching
            # them from here would race with the caller's own `catch_warnings` blocks)
            self._first = next(self._gen)
        except StopIteration:
            self._end = True
        except BaseException as e:  # noqa: BLE001 - re-raised by the iterator
            self._err = e

    def __iter__(self):
        self._thread.join()
        if self._err is not None:
            raise self._err
        if self._end:
            return
        yield self._first
        yield from self._gen

    def close(self):
        self._thread.join()
        self._gen.close()



class ContigFeed:
    """One process, one file, every contig it needs: the decode runs ahead on a helper thread (``EarlyContigs``) from
    the moment the command starts, and the command's per-contig work (a kernel launch, a handful of result rows) is done
    as each contig becomes resident - while the decoder is already in the next one - instead of after ``open_source``
    has decoded the whole file.  Iterate for ``(src, contig)`` in the order the contigs become resident; ``finish()``
    drains what is left and returns the source.  ``names``: the contigs the caller can want (``None``: all) - a lazily
    indexed file is then read through its index for those alone."""

    def __init__(self, input_file, workers=None, names=None, warn_bed6: bool = True):
        # (the decoder may run a whole genome's contigs ahead of a consumer that is busy with one contig's rows: the
        #  finished tables wait in HBM, 10 B per fragment)
        self._early = EarlyContigs(input_file, workers, True, warn_bed6, names=names, queued=32)
        self._it = None
        self.src: Optional[FragSource] = None
        self.seen: list = []
        self._input, self._workers, self._warn = input_file, workers, warn_bed6

    def __iter__(self):
        if self._it is None:
            self._it = iter(self._early)
        for src, c in self._it:
            self.src = src
            self.seen.append(c)
            yield src, c

    def require(self, contig: str) -> str:
        """Engine key of ``contig``, waiting for the decoder to reach it (contigs that become resident on the way
        stay resident) - the ``FragSource.require`` face of a feed, for consumers that ask contig by contig."""
        if contig not in self.seen:
            for _, c in self:
                if c == contig:
                    break
        if contig in self.seen:
            return self.src.key(contig)
        return self.finish().require(contig)  # not in the file: ValueError, as from a source

    def finish(self) -> FragSource:
        for _ in self:
            pass
        if self.src is None:  # a file without a single wanted contig: still a valid, open source
            self.src = open_source(self._input, self._workers, self._warn)
        return self.src

    def close(self):
        self._early.close()

def region_contig(input_file, contig: str, start: int, stop: int, workers: int | None = None, warn_bed6: bool = True):
    """``(src, key)``: the source of ``input_file`` and the engine key of a table with every fragment of ``contig``
    that overlaps ``[start, stop)`` (``FragSource.require_region``) - the way in for a rank that owns PART of a
    contig."""
    src = open_source(input_file, workers, warn_bed6)
    return src, src.require_region(contig, start, stop)


def _warn_bed6():
    # io/alignment.py:148-154
    warnings.warn(
        "input_file does not follow Fragmentation file format accepted by FinaleToolkit. "
        "Attempting to read as a BED6 file.", UserWarning)


def close_all():
    """Drop every cached source and the engine (tests / long-running hosts).  The HBM of the contigs is released;
    the engine's SCRATCH (up to 2.5 GB after a chr1 WPS) is parked in the library's cache of idle device blocks, where
    the next engine finds it - ``release_caches()`` gives that back to the device as well."""
    global _ENGINE
    for src in list(_SOURCES.values()):
        try:
            src.release()
        except Exception:
            pass
    _SOURCES.clear()
    global _SIDE_ENGINE
    if _SIDE_ENGINE is not None:
        _SIDE_ENGINE.close()
        _SIDE_ENGINE = None
    if _ENGINE is not None:
        _ENGINE.close()
        _ENGINE = None


def release_caches() -> int:
    """Give back what the library keeps between calls (idle page-locked blocks, idle device blocks, the streams'
    idle buffer sets: ``ftk_cache_trim``); returns the bytes released.  For long-lived hosts between jobs -
    nothing in use is touched, the next call allocates again."""
    from . import _lib
    return int(_lib.load().ftk_cache_trim())
