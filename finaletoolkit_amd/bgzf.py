"""
Minimal BGZF writer (blocked gzip, the container of ``.frag.gz`` / ``.bed.gz``
/ BAM files).  Used to materialise synthetic fragment files for tests and
benches; reading is done by the C++ decoder in ``csrc/ftk_decode.cpp``.
"""
from __future__ import annotations

import struct
import zlib

_BLOCK = 0xFF00  # uncompressed bytes per block (htslib's choice)
_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")


def _block(data: bytes, level: int) -> bytes:
    comp = zlib.compressobj(level, zlib.DEFLATED, -15)
    payload = comp.compress(data) + comp.flush()
    bsize = len(payload) + 25  # header 18 + trailer 8 - 1
    head = struct.pack("<BBBBIBBHBBHH", 31, 139, 8, 4, 0, 0, 255, 6, 66, 67, 2, bsize)
    tail = struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))
    return head + payload + tail


def write_bgzf(path, data: bytes, level: int = 6) -> list[int]:
    """Write ``data`` as a BGZF file with the standard empty EOF block; returns the file offset of every
    data block (block ``k`` holds bytes ``[k * 0xFF00, (k + 1) * 0xFF00)`` of ``data``)."""
    offsets = []
    with open(path, "wb") as fh:
        for off in range(0, len(data), _BLOCK):
            offsets.append(fh.tell())
            fh.write(_block(data[off:off + _BLOCK], level))
        offsets.append(fh.tell())  # the EOF block
        fh.write(_EOF)
    return offsets


def virtual_offset(block_offsets: list[int], byte_pos: int) -> int:
    """BGZF virtual offset of uncompressed byte ``byte_pos`` of a ``write_bgzf`` file."""
    k, u = divmod(byte_pos, _BLOCK)
    if k >= len(block_offsets) - 1:  # end of data: the EOF block
        return block_offsets[-1] << 16
    return (block_offsets[k] << 16) | u


def row_lengths(name: str, start, end, mapq, bed6: bool = False):
    """Bytes of every row ``name start end [. ]mapq strand`` of a fragment file (decimal digits counted, not
    formatted): what ``linear_index`` needs to place the rows of a contig in its text."""
    import numpy as np

    def digits(x):
        x = np.asarray(x, dtype=np.int64)
        d = np.ones(len(x), np.int64)
        for p in (10, 100, 1_000, 10_000, 100_000, 1_000_000, 10_000_000, 100_000_000, 1_000_000_000):
            d += x >= p
        return d
    return len(name.encode()) + 1 + digits(start) + 1 + digits(end) + 1 + (2 if bed6 else 0) + digits(mapq) + 1 + 1 + 1


def linear_index(start, end, row_bytes, block_offsets, first_byte: int = 0):
    """tabix linear index of one contig's rows (sorted by start): for every 16 kb window the virtual offset of the
    first row that overlaps it; a window without rows takes the next window's value, as htslib writes it.  Rows lie
    back to back from uncompressed byte ``first_byte`` of a ``write_bgzf``-shaped stream (0xFF00 bytes per block)
    whose blocks start at ``block_offsets``."""
    import numpy as np
    start = np.asarray(start, dtype=np.int64)
    end = np.asarray(end, dtype=np.int64)
    if len(start) == 0:
        return np.zeros(0, np.uint64)
    pos = first_byte + np.concatenate(([0], np.cumsum(np.asarray(row_bytes, dtype=np.int64))[:-1]))
    blk, within = np.divmod(pos, _BLOCK)
    voff = (np.asarray(block_offsets, dtype=np.int64)[blk].astype(np.uint64) << np.uint64(16)) | within.astype(np.uint64)
    n_win = int((max(int(end.max()), 1) - 1) >> 14) + 1
    w0 = np.arange(n_win, dtype=np.int64) << 14
    # first row (file order) with end > window start; it overlaps the window iff it starts before the window's end
    first = np.searchsorted(np.maximum.accumulate(end), w0, side="right")
    ok = first < len(start)
    ok[ok] &= start[first[ok]] < w0[ok] + (1 << 14)
    out = np.zeros(n_win, np.uint64)
    out[ok] = voff[first[ok]]
    nxt = np.uint64(0)
    for w in range(n_win - 1, -1, -1):  # (htslib: an empty window points at the next one's rows)
        if ok[w]:
            nxt = out[w]
        else:
            out[w] = nxt
    return out


def write_index(path, bai: bool, spans: list[tuple[str, int, int]], linear=None) -> None:
    """Minimal tabix (``bai=False``, BGZF-compressed) / BAI index: per reference one pseudo-bin 37450 whose
    first chunk is the reference's virtual-offset span ``(name, v_begin, v_end)`` -- what a reader needs to
    find a contig; ``v_begin == v_end`` marks a reference without records.  ``linear``: per reference the 16 kb
    linear index (``linear_index``) or None -- what a reader needs to start inside a contig."""
    body = b""
    for k, (_, vb, ve) in enumerate(spans):
        if ve > vb:
            body += struct.pack("<i", 1) + struct.pack("<Ii", 37450, 2) + struct.pack("<QQQQ", vb, ve, 0, 0)
        else:
            body += struct.pack("<i", 0)
        lin = None if linear is None else linear[k]
        if lin is None or len(lin) == 0:
            body += struct.pack("<i", 0)  # no linear index
        else:
            import numpy as np
            body += struct.pack("<i", len(lin)) + np.asarray(lin, dtype="<u8").tobytes()
    if bai:
        with open(path, "wb") as fh:
            fh.write(b"BAI\1" + struct.pack("<i", len(spans)) + body)
        return
    names = b"".join(n.encode() + b"\0" for n, _, _ in spans)
    head = b"TBI\1" + struct.pack("<iiiiiiii", len(spans), 0x10000, 1, 2, 3, ord("#"), 0, len(names)) + names
    write_bgzf(path, head + body)


def write_frag_gz(path, contig_rows, bed6: bool = False, with_tbi_stub: bool = True, level: int = 6,
                  with_index: bool = False) -> None:
    """Write a FinaleDB fragment file (``chrom start stop mapq strand``; BED6
    inserts a ``.`` name column).  ``contig_rows`` is an iterable of
    ``(name, start[], end[], mapq[], strand[])`` in file order.

    ``with_index`` writes a minimal tabix index (contig spans) next to it, which lets the engine decode
    single contigs without reading the rest; otherwise ``with_tbi_stub`` drops an empty ``<path>.tbi``
    (the engine then scans the file, but the reference's "index must exist" check,
    io/alignment.py:191-201, is kept).
    """
    contig_rows = list(contig_rows)
    if sum(len(r[1]) for r in contig_rows) > 50_000:
        write_frag_gz_contigs(path, contig_rows, bed6, level, with_index, with_tbi_stub)
        return
    parts = []
    marks = []  # (name, first byte, end byte) per run of a contig
    pos = 0
    for name, start, end, mapq, strand in contig_rows:
        first = pos
        for s, e, q, st in zip(start, end, mapq, strand):
            sign = "+" if st else "-"
            if bed6:
                parts.append(f"{name}\t{int(s)}\t{int(e)}\t.\t{int(q)}\t{sign}\n")
            else:
                parts.append(f"{name}\t{int(s)}\t{int(e)}\t{int(q)}\t{sign}\n")
            pos += len(parts[-1])
        marks.append((name, first, pos))
    offsets = write_bgzf(path, "".join(parts).encode(), level)
    if with_index:
        linear = [linear_index(r[1], r[2], row_lengths(r[0], r[1], r[2], r[3], bed6), offsets, m[1])
                  for r, m in zip(contig_rows, marks)]
        write_index(str(path) + ".tbi", False,
                    [(n, virtual_offset(offsets, a), virtual_offset(offsets, b)) for n, a, b in marks], linear)
    elif with_tbi_stub:
        open(str(path) + ".tbi", "ab").close()


def write_frag_gz_contigs(path, contig_rows, bed6: bool = False, level: int = 6, with_index: bool = True,
                          with_tbi_stub: bool = True) -> dict:
    """The large-file form of ``write_frag_gz``: ``contig_rows`` may be a GENERATOR - one contig's columns are alive at
    a time, so a file larger than memory (or than 4 GiB) can be written.  Rows are formatted and BGZF blocks deflated
    by the library's host threads, one contig after the other; a contig starts on a fresh block, so its virtual
    offset is just the block's file offset.  Returns ``{name: dict(first_off, end_off, linear)}`` (file offsets of
    the contig's first block and behind its last; its tabix linear index or None)."""
    from . import writers
    spans, linear, out = [], [], {}
    first = True
    for name, start, end, mapq, strand in contig_rows:
        with writers.frag_rows(name, start, end, mapq, strand, bed6) as rows:
            offs = writers.bgzf_write(path, rows, level, append=not first, write_eof=False)
        spans.append((name, int(offs[0]), int(offs[-1])))
        lin = linear_index(start, end, row_lengths(name, start, end, mapq, bed6), offs) if with_index else None
        linear.append(lin)
        out[name] = dict(first_off=int(offs[0]), end_off=int(offs[-1]), linear=lin)
        first = False
    with open(path, "ab" if not first else "wb") as fh:
        fh.write(_EOF)
    if with_index:
        write_index(str(path) + ".tbi", False, [(n, a << 16, b << 16) for n, a, b in spans], linear)
    elif with_tbi_stub:
        open(str(path) + ".tbi", "ab").close()
    return out
