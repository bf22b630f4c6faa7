"""Shared test helpers (independent of the product decoder)."""
import gzip
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DATA = os.path.join(ROOT, "tests", "data")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def read_frag_gz(path):
    """Parse a FinaleDB / BED6 fragment file with Python's gzip: {contig: (start, end, mapq, strand)}."""
    rows = {}
    bed6 = None
    with gzip.open(path, "rt") as fh:
        for line in fh:
            if not line.strip() or line.startswith("#"):
                continue
            p = line.rstrip("\n").split("\t")
            if bed6 is None:
                bed6 = len(p) > 5
            mq, st = (p[4], p[5]) if bed6 else (p[3], p[4])
            rows.setdefault(p[0], []).append((int(p[1]), int(p[2]), int(mq), 1 if "+" in st else 0))
    out = {}
    for c, r in rows.items():
        a = np.array(r, dtype=np.int64)
        out[c] = (a[:, 0].astype(np.int32), a[:, 1].astype(np.int32), a[:, 2].astype(np.uint8),
                  a[:, 3].astype(np.uint8))
    return out


def golden_json():
    with open(os.path.join(GOLDEN, "golden.json")) as fh:
        return json.load(fh)


def golden_npz():
    return np.load(os.path.join(GOLDEN, "golden.npz"))


def read_bed(path):
    """BED reader with the reference's skipping rules (utils/utils.py:310-343)."""
    out = []
    with open(path) as fh:
        for line in fh:
            if line.startswith(("#", "track", "browser")) or not line.strip():
                continue
            p = line.strip().split("\t")
            if len(p) < 3:
                continue
            out.append((p[0], int(p[1]), int(p[2]), p[3] if len(p) > 3 else "."))
    return out


def write_synthetic_bam(path, contigs, frags, junk=True, read_len=100, index=True):
    """Write a coordinate-sorted BAM (+ empty .bai stub) holding one read pair per fragment.

    contigs: [(name, length)]; frags: {name: (start[], end[], mapq[], forward[])}.  Forward fragments put
    read1 at the fragment start (flag 99 / mate 147), reverse ones at its end (flag 83 / mate 163).  With
    ``junk`` a few records the reference must drop are mixed in (unmapped, secondary, duplicate, qc-fail,
    supplementary, improper pair, mate unmapped, unpaired, TLEN 0).  Returns the expected per-contig rows
    ``(start, end, mapq, forward, r1_start, r1_end, file_rank)`` in the decoder's order (stable sort by
    fragment start of the read1 records in file order); ``file_rank`` restores the file order."""
    import struct
    from finaletoolkit_amd import bgzf

    def rec(ref_id, pos, mapq, flag, tlen, name, rl, mate_pos):
        nm = name.encode() + b"\0"
        cigar = struct.pack("<I", (rl << 4) | 0) if rl > 0 else b""
        body = struct.pack("<iiBBHHHiiii", ref_id, pos, len(nm), mapq, 4680, 1 if rl > 0 else 0, flag, rl, ref_id,
                           mate_pos, tlen) + nm + cigar + b"\x11" * ((rl + 1) // 2) + b"\xff" * rl
        return struct.pack("<i", len(body)) + body

    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{c}\tLN:{n}\n" for c, n in contigs)
    out = [b"BAM\1", struct.pack("<i", len(text)), text.encode(), struct.pack("<i", len(contigs))]
    for c, n in contigs:
        out.append(struct.pack("<i", len(c) + 1) + c.encode() + b"\0" + struct.pack("<i", n))
    expected = {}
    spans = []  # byte range of every contig's records in the uncompressed stream
    for ref_id, (c, _) in enumerate(contigs):
        s, e, q, fw = frags.get(c, ([], [], [], []))
        records = []  # (pos, order, bytes, read1 fragment or None)
        k = 0
        for i in range(len(s)):
            fs, fe, mq, f = int(s[i]), int(e[i]), int(q[i]), bool(fw[i])
            ln = fe - fs
            rl = min(read_len, ln)
            if f:
                records.append((fs, k, rec(ref_id, fs, mq, 99, ln, f"f{i}", rl, fe - rl), (fs, fe, mq, 1, fs, fs + rl)))
                records.append((fe - rl, k + 1, rec(ref_id, fe - rl, mq, 147, -ln, f"f{i}", rl, fs), None))
            else:
                records.append((fe - rl, k, rec(ref_id, fe - rl, mq, 83, -ln, f"f{i}", rl, fs), (fs, fe, mq, 0, fe - rl, fe)))
                records.append((fs, k + 1, rec(ref_id, fs, mq, 163, ln, f"f{i}", rl, fe - rl), None))
            k += 2
            if junk and i % 7 == 0:
                for flag, tl in ((99 | 0x400, ln), (99 | 0x100, ln), (99 | 0x200, ln), (99 | 0x800, ln), (97, ln),
                                 (99 | 0x8, ln), (0x40, ln), (99, 0), (0x4 | 0x41, 0)):
                    records.append((fs, k, rec(ref_id, fs, mq, flag, tl, f"j{i}", rl, fs), None))
                    k += 1
        records.sort(key=lambda r: (r[0], r[1]))
        first = sum(len(x) for x in out)
        out += [r[2] for r in records]
        spans.append((c, first, sum(len(x) for x in out)))
        rows = [r[3] for r in records if r[3] is not None]
        order = sorted(range(len(rows)), key=lambda j: rows[j][0])  # stable, like the decoder
        # 7th column: rank of the read1 record in the file (the order pysam iterates in)
        expected[c] = [rows[j] + (j,) for j in order]
    offsets = bgzf.write_bgzf(path, b"".join(out), level=1)
    if index:  # a real (minimal) BAI: lets the decoder jump to a contig
        bgzf.write_index(str(path) + ".bai", True, [(c, bgzf.virtual_offset(offsets, a), bgzf.virtual_offset(offsets, b))
                                                     for c, a, b in spans])
    else:
        open(str(path) + ".bai", "ab").close()
    return expected


def read_fasta_gz(path):
    """{contig: sequence} of a gzipped FASTA fixture."""
    seqs, name, parts = {}, None, []
    with gzip.open(path, "rt") as fh:
        for line in fh:
            if line.startswith(">"):
                if name is not None:
                    seqs[name] = "".join(parts)
                name, parts = line[1:].split()[0], []
            else:
                parts.append(line.strip())
    if name is not None:
        seqs[name] = "".join(parts)
    return seqs


def write_fasta(path, seqs, width=60, fai=True):
    """Plain FASTA (+ .fai) from {contig: sequence}."""
    index = []
    with open(path, "wb") as fh:
        for name, s in seqs.items():
            fh.write(f">{name}\n".encode())
            off = fh.tell()
            for i in range(0, len(s), width):
                fh.write(s[i:i + width].encode() + b"\n")
            index.append((name, len(s), off, width, width + 1))
    if fai:
        with open(str(path) + ".fai", "w") as fh:
            for row in index:
                fh.write("\t".join(map(str, row)) + "\n")


def write_2bit(path, seqs):
    """UCSC .2bit (little-endian, version 0) from {contig: sequence}: N runs -> N blocks,
    lower-case runs -> mask blocks, bases packed T=0 C=1 A=2 G=3 (N packed as T)."""
    import struct

    def runs(mask):
        d = np.diff(np.concatenate([[0], mask.astype(np.int8), [0]]))
        return np.flatnonzero(d == 1), np.flatnonzero(d == -1)

    records = []
    for name, s in seqs.items():
        b = np.frombuffer(s.encode(), np.uint8)
        up = b & 0xDF
        ns, ne = runs(up == ord("N"))
        ms, me = runs((b & 0x20) != 0)
        code = np.zeros(len(b), np.uint8)
        code[up == ord("C")] = 1
        code[up == ord("A")] = 2
        code[up == ord("G")] = 3
        pad = np.zeros((-len(b)) % 4, np.uint8)
        c4 = np.concatenate([code, pad]).reshape(-1, 4)
        packed = (c4[:, 0] << 6 | c4[:, 1] << 4 | c4[:, 2] << 2 | c4[:, 3]).astype(np.uint8)
        rec = struct.pack("<II", len(b), len(ns)) + ns.astype("<u4").tobytes() + (ne - ns).astype("<u4").tobytes()
        rec += struct.pack("<I", len(ms)) + ms.astype("<u4").tobytes() + (me - ms).astype("<u4").tobytes()
        rec += struct.pack("<I", 0) + packed.tobytes()
        records.append((name, rec))
    head = struct.pack("<IIII", 0x1A412743, 0, len(records), 0)
    index_len = sum(1 + len(n) + 4 for n, _ in records)
    off = len(head) + index_len
    index = b""
    for name, rec in records:
        index += bytes([len(name)]) + name.encode() + struct.pack("<I", off)
        off += len(rec)
    with open(path, "wb") as fh:
        fh.write(head + index + b"".join(r for _, r in records))


def synth_reference(path, contigs, seed=20261002):
    """Seeded synthetic reference genome as FASTA (+ .fai) for ``contigs = {name: length}``: random bases with
    GC-rich and GC-poor stretches, soft-masked (lower-case) runs and N runs.  Returns the sha256 of the FASTA
    bytes, which ``tests/golden/delfi_driver.json`` pins (the goldens were produced by the reference on exactly
    this file; it is regenerated instead of committed)."""
    import hashlib
    seqs = {}
    for i, (name, n) in enumerate(contigs.items()):
        rng = np.random.default_rng(seed + i)
        # piecewise GC content: blocks of 2-20 kb with their own G+C probability
        gc = np.empty(n, np.float64)
        pos = 0
        while pos < n:
            ln = int(rng.integers(2_000, 20_000))
            gc[pos:pos + ln] = rng.uniform(0.25, 0.65)
            pos += ln
        is_gc = rng.random(n) < gc
        pick = rng.integers(0, 2, n)
        codes = np.where(is_gc, np.where(pick == 1, ord("G"), ord("C")), np.where(pick == 1, ord("A"), ord("T"))).astype(np.uint8)
        for _ in range(6):  # N runs
            a = int(rng.integers(0, n - 3000))
            codes[a:a + int(rng.integers(50, 3000))] = ord("N")
        for _ in range(25):  # soft-masked runs
            a = int(rng.integers(0, n - 2000))
            b = a + int(rng.integers(100, 2000))
            codes[a:b] |= 0x20
        seqs[name] = codes.tobytes().decode()
    write_fasta(path, seqs)
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


def write_bins20(path, contigs):
    """The 20 bp bin tiling of the DELFI driver goldens (27 500 rows for the synthetic genome): a comment line, then
    chrB BEFORE chrA -- the driver orders by chrom.sizes, not by the bins file."""
    with open(path, "w") as fh:
        fh.write("# 20 bp tiling\n")
        for c in sorted(contigs, reverse=True):
            for a in range(0, contigs[c], 20):
                fh.write(f"{c}\t{a}\t{min(a + 19, contigs[c])}\n")


def bam_reg2bin(beg, end):
    """The SAM specification's ``reg2bin`` (the ``bin`` field of a BAM record)."""
    end -= 1
    for shift, base in ((14, 4681), (17, 585), (20, 73), (23, 9), (26, 1)):
        if beg >> shift == end >> shift:
            return base + (beg >> shift)
    return 0


_CIGAR_OPS = "MIDNSHP=XB"


def bam_record(ref_id, pos, mapq, flag, cigar, tlen, name, l_seq=None, mate_ref=None, mate_pos=0, aux=b""):
    """One BAM alignment record (block_size included).  ``cigar``: text (``"5S40M3I2D"``, ``"*"`` / ``""`` = none) or
    ``[(op, length)]`` with numeric ops; ``l_seq`` defaults to the query length the CIGAR implies."""
    import re
    import struct
    if isinstance(cigar, str):
        cigar = [] if cigar in ("", "*") else [(_CIGAR_OPS.index(o), int(n)) for n, o in re.findall(r"(\d+)([MIDNSHP=XB])", cigar)]
    if l_seq is None:
        l_seq = sum(n for op, n in cigar if op in (0, 1, 4, 7, 8))
    rlen = 0 if flag & 0x4 else sum(n for op, n in cigar if op in (0, 2, 3, 7, 8))
    nm = name.encode() + b"\0"
    body = struct.pack("<iiBBHHHiiii", ref_id, pos, len(nm), mapq, bam_reg2bin(max(pos, 0), max(pos, 0) + (rlen or 1)),
                       len(cigar), flag, l_seq, ref_id if mate_ref is None else mate_ref, mate_pos, tlen)
    body += nm + b"".join(struct.pack("<I", (n << 4) | op) for op, n in cigar)
    body += b"\x12" * ((l_seq + 1) // 2) + b"\x1e" * l_seq + aux
    return struct.pack("<i", len(body)) + body


def write_bam(path, contigs, records, level=6):
    """Write ``records`` -- ``[(ref_id, pos, bytes from bam_record)]``, already in file order -- as a BAM with the
    header of ``contigs = [(name, length)]`` and a minimal BAI (every reference's span), using only ``zlib``."""
    import struct
    from finaletoolkit_amd import bgzf
    text = "@HD\tVN:1.6\tSO:coordinate\n" + "".join(f"@SQ\tSN:{c}\tLN:{n}\n" for c, n in contigs)
    head = b"BAM\1" + struct.pack("<i", len(text)) + text.encode() + struct.pack("<i", len(contigs))
    for c, n in contigs:
        head += struct.pack("<i", len(c) + 1) + c.encode() + b"\0" + struct.pack("<i", n)
    spans = {}
    o = len(head)
    for ref_id, _, b in records:
        a, _ = spans.get(ref_id, (o, o))
        spans[ref_id] = (a, o + len(b))
        o += len(b)
    offsets = bgzf.write_bgzf(path, head + b"".join(r[2] for r in records), level=level)
    bgzf.write_index(str(path) + ".bai", True,
                     [(c, bgzf.virtual_offset(offsets, spans.get(k, (0, 0))[0]), bgzf.virtual_offset(offsets, spans.get(k, (0, 0))[1]))
                      for k, (c, _) in enumerate(contigs)])
