"""
Reference-genome access for DELFI's per-bin GC fraction
(``frag/_delfi.py:476-490`` of the reference, which goes through
``io/reference.py:120-189``): a dependency-free reader for UCSC ``.2bit`` and
(faidx-indexed or plain) FASTA files exposing ``chroms`` and a vectorised
``gc_count(contig, start, stop)`` = number of G + C bases (case-insensitive),
i.e. ``seq.upper().count("G") + seq.upper().count("C")``.
"""
from __future__ import annotations

import os
import struct

import numpy as np

_TWOBIT_SUFFIXES = (".2bit", ".tb2")
_SIG = 0x1A412743
# 2-bit codes: T=0, C=1, A=2, G=3 -> G/C are the codes with the low bit set
_GC_PER_BYTE = np.array([bin(b & 0x55).count("1") for b in range(256)], dtype=np.uint8)


class ReferenceGenome:
    def __init__(self, reference_path):
        self.path = str(reference_path)
        if not os.path.exists(self.path):
            raise FileNotFoundError(f"Reference file not found: {self.path}")
        self.is_2bit = self.path.endswith(_TWOBIT_SUFFIXES)
        self._fh = open(self.path, "rb")
        if self.is_2bit:
            self._open_2bit()
        else:
            self._open_fasta()

    # -- 2bit -------------------------------------------------------------------
    def _open_2bit(self):
        f = self._fh
        head = f.read(16)
        sig = struct.unpack("<I", head[:4])[0]
        self._end = "<"
        if sig != _SIG:
            if struct.unpack(">I", head[:4])[0] != _SIG:
                raise ValueError(f"{self.path} is not a 2bit file")
            self._end = ">"
        _, version, count, _ = struct.unpack(self._end + "IIII", head)
        if version != 0:
            raise ValueError("unsupported 2bit version")
        self._offsets = {}
        for _ in range(count):
            n = f.read(1)[0]
            name = f.read(n).decode()
            self._offsets[name] = struct.unpack(self._end + "I", f.read(4))[0]
        self._records = {}
        self.chroms = {}
        for name, off in self._offsets.items():
            f.seek(off)
            size, nb = struct.unpack(self._end + "II", f.read(8))
            n_starts = np.frombuffer(f.read(4 * nb), dtype=self._end + "u4").astype(np.int64)
            n_sizes = np.frombuffer(f.read(4 * nb), dtype=self._end + "u4").astype(np.int64)
            mb = struct.unpack(self._end + "I", f.read(4))[0]
            f.seek(8 * mb + 4, 1)  # soft-mask blocks + reserved
            self._records[name] = (size, n_starts, n_sizes, f.tell())
            self.chroms[name] = size

    def _gc_2bit(self, contig, start, stop):
        size, n_starts, n_sizes, dna_off = self._records[contig]
        b0, b1 = start // 4, (stop + 3) // 4
        self._fh.seek(dna_off + b0)
        packed = np.frombuffer(self._fh.read(b1 - b0), dtype=np.uint8)
        codes = np.empty(len(packed) * 4, np.uint8)
        codes[0::4] = packed >> 6
        codes[1::4] = (packed >> 4) & 3
        codes[2::4] = (packed >> 2) & 3
        codes[3::4] = packed & 3
        codes = codes[start - 4 * b0: stop - 4 * b0]
        gc = (codes & 1).astype(bool)
        for s, n in zip(n_starts, n_sizes):  # N blocks count as neither
            a, b = max(s, start), min(s + n, stop)
            if a < b:
                gc[a - start:b - start] = False
        return int(gc.sum())

    # -- FASTA --------------------------------------------------------------------
    def _open_fasta(self):
        if self.path.endswith(".gz"):
            raise ValueError("compressed FASTA is not supported; decompress it or use a .2bit reference")
        self._fai = {}
        fai = self.path + ".fai"
        if os.path.exists(fai):
            with open(fai) as fh:
                for line in fh:
                    p = line.rstrip("\n").split("\t")
                    if len(p) >= 5:
                        self._fai[p[0]] = (int(p[1]), int(p[2]), int(p[3]), int(p[4]))
        else:  # index in memory (the reference would run pysam.faidx here)
            f = self._fh
            f.seek(0)
            name = None
            pos = 0
            length = offset = linebases = linewidth = 0
            for raw in f:
                if raw.startswith(b">"):
                    if name is not None:
                        self._fai[name] = (length, offset, linebases, linewidth)
                    name = raw[1:].split()[0].decode()
                    length = linebases = linewidth = 0
                    offset = pos + len(raw)
                else:
                    stripped = raw.rstrip(b"\r\n")
                    if linebases == 0:
                        linebases, linewidth = len(stripped), len(raw)
                    length += len(stripped)
                pos += len(raw)
            if name is not None:
                self._fai[name] = (length, offset, linebases, linewidth)
        self.chroms = {k: v[0] for k, v in self._fai.items()}

    def _gc_fasta(self, contig, start, stop):
        length, offset, linebases, linewidth = self._fai[contig]
        if linebases == 0:
            return 0
        o0 = offset + (start // linebases) * linewidth + start % linebases
        o1 = offset + (stop // linebases) * linewidth + stop % linebases
        self._fh.seek(o0)
        raw = np.frombuffer(self._fh.read(o1 - o0), dtype=np.uint8)
        return int(np.isin(raw, np.frombuffer(b"GCgc", dtype=np.uint8)).sum())

    # -- public ---------------------------------------------------------------------
    def gc_count(self, contig: str, start: int, stop: int) -> int:
        """G + C bases in ``contig:[start, stop)``; caller validates the interval."""
        if stop <= start:
            return 0
        return self._gc_2bit(contig, start, stop) if self.is_2bit else self._gc_fasta(contig, start, stop)

    def gc_counts(self, engine, contig: str, starts, stops) -> np.ndarray:
        """G + C per interval ``[starts[i], stops[i])`` of one contig, counted on the device from a
        reference image uploaded once per contig (``ftk_ref_upload`` / ``ftk_ref_gc_counts``).
        Same numbers as ``gc_count`` (which stays as the host-side definition used by the tests)."""
        starts = np.asarray(starts, dtype=np.int64)
        stops = np.asarray(stops, dtype=np.int64)
        rid = self.device_image(engine, contig, with_layout=False)
        if self.is_2bit:
            size, n_starts, n_sizes, dna_off = self._records[contig]
            counts = engine.ref_gc_counts(rid, starts, stops)
            # bases inside N blocks count as neither (they are stored as some code): subtract their share
            for s0, n in zip(n_starts, n_sizes):
                a = np.maximum(starts, s0)
                b = np.minimum(stops, s0 + n)
                hit = np.nonzero(a < b)[0]
                if len(hit):
                    counts[hit] -= engine.ref_gc_counts(rid, a[hit], b[hit])
            return counts
        length, offset, linebases, linewidth = self._fai[contig]
        if linebases == 0:
            return np.zeros(len(starts), np.int64)
        lo = (starts // linebases) * linewidth + starts % linebases
        hi = (stops // linebases) * linewidth + stops % linebases
        return engine.ref_gc_counts(rid, lo, hi)

    def device_image(self, engine, contig: str, with_layout: bool = True) -> int:
        """Upload ``contig``'s sequence image once (cached on the engine) -- the packed DNA of a 2bit record or the
        raw FASTA text, read from the file by the library (``ftk_ref_upload_file``) -- and return the reference id
        the device calls take.  ``with_layout`` also hands over its geometry (N blocks / FASTA line layout), which
        the motif kernels need and the GC count does not."""
        key = (self.path, contig)
        rid = engine.__dict__.get("_refs", {}).get(key)
        done = engine.__dict__.setdefault("_ref_layouts", set())
        if rid is not None and (not with_layout or rid in done):
            return rid
        if self.is_2bit:
            size, n_starts, n_sizes, dna_off = self._records[contig]
            if rid is None:
                rid = engine.ref_upload_file(key, self.path, dna_off, (size + 3) // 4, 1)
                done.discard(rid)
            if with_layout:
                order = np.argsort(n_starts, kind="stable")
                engine.ref_set_layout(rid, size, 0, 0, n_starts[order], (n_starts + n_sizes)[order])
                done.add(rid)
            return rid
        length, offset, linebases, linewidth = self._fai[contig]
        if linebases == 0:
            if rid is None:
                rid = engine.ref_upload(key, np.zeros(0, np.uint8), 0)
            engine.ref_set_layout(rid, 0, 1, 1)
            done.add(rid)
            return rid
        n_text = (length // linebases) * linewidth + length % linebases
        if rid is None:
            rid = engine.ref_upload_file(key, self.path, offset, n_text, 0)
            done.discard(rid)
        if with_layout:
            engine.ref_set_layout(rid, length, linebases, linewidth)
            done.add(rid)
        return rid

    def close(self):
        if self._fh:
            self._fh.close()
            self._fh = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
