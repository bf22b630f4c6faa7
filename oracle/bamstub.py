"""
TEST INFRASTRUCTURE ONLY -- never imported by the product path.

``pysam.AlignmentFile`` / ``pysam.AlignedSegment`` stand-ins for ``oracle/refstub.py``: a BAM file read with
Python's ``gzip`` (BGZF is a multi-member gzip file) and ``struct``, exposing exactly the attributes the reference's
BAM path touches (``/root/reference/src/finaletoolkit/io/alignment.py:60-71,124-127,182-188,242-268``), with the
semantics of pysam 0.24 / htslib 1.2x that the reference relies on.  With it the IMPORTED reference itself emits the
BAM-mode goldens (``oracle/gen_golden_bam.py``); nothing here restates the reference.

What is restated is third-party behaviour (pysam / htslib are absent from this image).  Each rule below names its source:

* ``AlignedSegment.reference_end`` -- pysam ``libcalignedsegment.pyx`` (property ``reference_end``): ``None`` when the
  read is unmapped (flag 0x4) **or has no CIGAR**, else htslib ``bam_endpos``.
* ``bam_endpos`` -- htslib ``sam.c``: ``pos + rlen`` with ``rlen = bam_cigar2rlen`` (0 for unmapped reads) and
  **``rlen == 0`` counted as 1**.
* ``bam_cigar2rlen`` -- htslib ``sam.h``: ops that consume the reference are ``M D N = X`` (``BAM_CIGAR_TYPE`` bit 1).
* ``AlignmentFile.fetch(contig, start, stop)`` -- pysam ``libcalignmentfile.pyx`` (``fetch`` / ``parse_region``) over
  htslib's iterator (``hts.c``, ``hts_itr_next``): needs an index; no contig = every reference in header order
  (``IteratorRowAllRefs``: records without a reference are not returned) and start/stop are ignored; with a contig the
  records of that reference with ``pos < stop and bam_endpos > start`` in file order; unknown contig, ``start < 0``,
  ``start > stop`` raise ``ValueError``.
* flag properties -- the SAM flag bits (``is_paired`` 0x1, ``is_proper_pair`` 0x2, ``is_unmapped`` 0x4,
  ``mate_is_unmapped`` 0x8, ``is_reverse`` 0x10, ``is_read1`` 0x40, ``is_read2`` 0x80, ``is_secondary`` 0x100,
  ``is_qcfail`` 0x200, ``is_duplicate`` 0x400, ``is_supplementary`` 0x800); ``is_forward = not is_reverse``.
"""
from __future__ import annotations

import gzip
import os
import struct

MAX_POS = (1 << 31) - 1
_REF_OPS = (0, 2, 3, 7, 8)  # M D N = X


class AlignedSegment:
    __slots__ = ("_file", "reference_id", "reference_start", "mapping_quality", "flag", "template_length",
                 "next_reference_id", "next_reference_start", "query_name", "cigartuples", "query_length", "_endpos")

    def __init__(self, file, body):
        (ref_id, pos, l_name, mapq, _bin, n_cigar, flag, l_seq, next_ref, next_pos,
         tlen) = struct.unpack_from("<iiBBHHHiiii", body, 0)
        self._file = file
        self.reference_id = ref_id
        self.reference_start = pos
        self.mapping_quality = mapq
        self.flag = flag
        self.template_length = tlen
        self.next_reference_id = next_ref
        self.next_reference_start = next_pos
        self.query_name = body[32:32 + l_name - 1].decode()
        ops = struct.unpack_from(f"<{n_cigar}I", body, 32 + l_name)
        self.cigartuples = [(v & 15, v >> 4) for v in ops]
        self.query_length = l_seq
        # htslib bam_endpos: unmapped reads and alignments that consume no reference count as one base
        rlen = 0 if (flag & 0x4) else sum(n for op, n in self.cigartuples if op in _REF_OPS)
        self._endpos = pos + (rlen or 1)

    # --- flags
    is_paired = property(lambda s: bool(s.flag & 0x1))
    is_proper_pair = property(lambda s: bool(s.flag & 0x2))
    is_unmapped = property(lambda s: bool(s.flag & 0x4))
    mate_is_unmapped = property(lambda s: bool(s.flag & 0x8))
    is_reverse = property(lambda s: bool(s.flag & 0x10))
    is_forward = property(lambda s: not (s.flag & 0x10))
    mate_is_reverse = property(lambda s: bool(s.flag & 0x20))
    is_read1 = property(lambda s: bool(s.flag & 0x40))
    is_read2 = property(lambda s: bool(s.flag & 0x80))
    is_secondary = property(lambda s: bool(s.flag & 0x100))
    is_qcfail = property(lambda s: bool(s.flag & 0x200))
    is_duplicate = property(lambda s: bool(s.flag & 0x400))
    is_supplementary = property(lambda s: bool(s.flag & 0x800))

    @property
    def reference_name(self):
        return None if self.reference_id < 0 else self._file.references[self.reference_id]

    @property
    def reference_end(self):
        if (self.flag & 0x4) or not self.cigartuples:
            return None
        return self._endpos


class AlignmentFile:
    def __init__(self, filename, mode="r", reference_filename=None, threads=1, **kwargs):
        self.filename = str(filename)
        if not self.filename.lower().endswith(".bam"):
            raise ValueError("the stand-in reads BAM only (no htslib here)")
        with gzip.open(self.filename, "rb") as fh:
            data = fh.read()
        if data[:4] != b"BAM\1":
            raise ValueError("not a BAM file")
        (l_text,) = struct.unpack_from("<i", data, 4)
        self.text = data[8:8 + l_text].decode(errors="replace")
        o = 8 + l_text
        (n_ref,) = struct.unpack_from("<i", data, o)
        o += 4
        names, lengths = [], []
        for _ in range(n_ref):
            (l_name,) = struct.unpack_from("<i", data, o)
            names.append(data[o + 4:o + 4 + l_name - 1].decode())
            (ln,) = struct.unpack_from("<i", data, o + 4 + l_name)
            lengths.append(ln)
            o += 8 + l_name
        self.references = tuple(names)
        self.lengths = tuple(lengths)
        self.nreferences = n_ref
        self._records = []
        while o + 4 <= len(data):
            (bs,) = struct.unpack_from("<i", data, o)
            self._records.append(AlignedSegment(self, data[o + 4:o + 4 + bs]))
            o += 4 + bs
        self._by_tid = {}
        for r in self._records:
            self._by_tid.setdefault(r.reference_id, []).append(r)
        self._has_index = any(os.path.exists(p) for p in (self.filename + ".bai", self.filename[:-4] + ".bai",
                                                          self.filename + ".csi"))

    def has_index(self):
        return self._has_index

    def get_tid(self, contig):
        try:
            return self.references.index(contig)
        except ValueError:
            return -1

    def get_reference_name(self, tid):
        return self.references[tid]

    def fetch(self, contig=None, start=None, stop=None, region=None, tid=None, until_eof=False,
              multiple_iterators=False, reference=None, end=None):
        if reference is not None:
            contig = reference
        if end is not None:
            stop = end
        if region is not None:
            raise NotImplementedError("region strings are not used by the reference's hot path")
        if until_eof:
            yield from self._records
            return
        if not self._has_index:
            raise ValueError("fetch called on bamfile without index")
        if contig is None and tid is None:
            # IteratorRowAllRefs: one index query per reference, in header order
            for t in range(self.nreferences):
                yield from self._by_tid.get(t, ())
            return
        rstart = 0 if start is None else int(start)
        rstop = MAX_POS if stop is None else int(stop)
        rtid = tid if tid is not None else self.get_tid(contig)
        if rtid < 0 or rtid >= self.nreferences:
            raise ValueError(f"invalid contig `{contig}`")
        if rstart > rstop:
            raise ValueError(f"invalid coordinates: start ({rstart}) > stop ({rstop})")
        if not 0 <= rstart < MAX_POS:
            raise ValueError(f"start out of range ({rstart})")
        if not 0 <= rstop <= MAX_POS:
            raise ValueError(f"stop out of range ({rstop})")
        for r in self._by_tid.get(rtid, ()):
            if r.reference_start < rstop and r._endpos > rstart:
                yield r

    def close(self):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False
