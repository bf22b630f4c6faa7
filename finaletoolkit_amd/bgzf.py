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


def write_bgzf(path, data: bytes, level: int = 6) -> None:
    """Write ``data`` as a BGZF file with the standard empty EOF block."""
    with open(path, "wb") as fh:
        for off in range(0, len(data), _BLOCK):
            fh.write(_block(data[off:off + _BLOCK], level))
        fh.write(_EOF)


def write_frag_gz(path, contig_rows, bed6: bool = False, with_tbi_stub: bool = True, level: int = 6) -> None:
    """Write a FinaleDB fragment file (``chrom start stop mapq strand``; BED6
    inserts a ``.`` name column).  ``contig_rows`` is an iterable of
    ``(name, start[], end[], mapq[], strand[])`` in file order.

    ``with_tbi_stub`` drops an empty ``<path>.tbi`` next to it: the engine
    decodes whole contigs and never reads the index, but keeps the reference's
    "index must exist" check (io/alignment.py:191-201).
    """
    parts = []
    for name, start, end, mapq, strand in contig_rows:
        for s, e, q, st in zip(start, end, mapq, strand):
            sign = "+" if st else "-"
            if bed6:
                parts.append(f"{name}\t{int(s)}\t{int(e)}\t.\t{int(q)}\t{sign}\n")
            else:
                parts.append(f"{name}\t{int(s)}\t{int(e)}\t{int(q)}\t{sign}\n")
    write_bgzf(path, "".join(parts).encode(), level)
    if with_tbi_stub:
        open(str(path) + ".tbi", "ab").close()
