"""
bigWig output for ``multi_wps`` without pyBigWig: the equivalent of
``bigwig.addHeader(header)`` + ``bigwig.addEntries(chrom, start, values=f64,
span=1, step=1)`` per interval (reference ``frag/_multi_wps.py:300-325``), i.e.
zlib-compressed **fixedStep** sections of float32 values, a chromosome B+ tree
and an R-tree index (bigWig v4, little endian).  One zoom level of per-section
summaries is written so browsers and ``stats`` readers that insist on a zoom
level have one.  ``read_bigwig`` is the matching minimal reader (tests, and a
way to inspect results without pyBigWig).
"""
from __future__ import annotations

import struct
import sys
import zlib

import numpy as np

_BW_MAGIC = 0x888FFC26
_CHROM_TREE_MAGIC = 0x78CA8C91
_RTREE_MAGIC = 0x2468ACE0
_ITEMS_PER_SECTION = 16384  # <= 65535 (u16 item count)
_BLOCK = 256                # children per index node
_ZOOM_SLOTS = 10            # zoom headers a file has room for (libBigWig reserves ten)


def _chrom_tree(header) -> bytes:
    """B+ tree over ``(name -> id, size)``; ids follow header order."""
    key = max([len(c.encode()) for c, _ in header] + [1])
    items = sorted(((c.encode().ljust(key, b"\0"), i, int(n)) for i, (c, n) in enumerate(header)), key=lambda t: t[0])
    out = struct.pack("<IIIIQQ", _CHROM_TREE_MAGIC, _BLOCK, key, 8, len(items), 0)
    base = 0  # node offsets are absolute file offsets; patched by the caller through `reloc`
    # levels bottom-up: leaves hold (key, id, size); inner nodes (key, child offset)
    leaves = [items[i:i + _BLOCK] for i in range(0, len(items), _BLOCK)] or [[]]
    levels = [leaves]
    while len(levels[-1]) > 1:
        prev = levels[-1]
        levels.append([prev[i:i + _BLOCK] for i in range(0, len(prev), _BLOCK)])
    levels.reverse()  # root first
    # sizes
    sizes = []
    for li, lv in enumerate(levels):
        leaf = li == len(levels) - 1
        sizes.append([4 + len(node) * (key + 8) for node in lv])
    offs = []
    pos = len(out)
    for lv_sizes in sizes:
        offs.append([])
        for sz in lv_sizes:
            offs[-1].append(pos)
            pos += sz
    body = bytearray()
    reloc = []  # positions (in `out + body`) of child offsets that need the tree's file offset added

    def first_key(node, li):
        while li < len(levels) - 1:
            node = node[0]
            li += 1
        return node[0][0] if node else b"\0" * key

    for li, lv in enumerate(levels):
        leaf = li == len(levels) - 1
        child = 0
        for node in lv:
            body += struct.pack("<BBH", 1 if leaf else 0, 0, len(node))
            for it in node:
                if leaf:
                    body += it[0] + struct.pack("<II", it[1], it[2])
                else:
                    body += first_key(it, li + 1)
                    reloc.append(len(out) + len(body))
                    body += struct.pack("<Q", offs[li + 1][child])
                    child += 1
    return out + bytes(body), reloc


def _rtree(leaf_items, index_offset) -> bytes:
    """R-tree over section records ``(chrom, start, chrom, end, offset, size)``."""
    n = len(leaf_items)
    if n:
        bounds = (leaf_items[0][0], leaf_items[0][1], leaf_items[-1][2], leaf_items[-1][3],
                  leaf_items[-1][4] + leaf_items[-1][5])
    else:
        bounds = (0, 0, 0, 0, index_offset)
    head = struct.pack("<IIQIIIIQII", _RTREE_MAGIC, _BLOCK, n, bounds[0], bounds[1], bounds[2], bounds[3], bounds[4],
                       1, 0)
    levels = [[leaf_items[i:i + _BLOCK] for i in range(0, n, _BLOCK)] or [[]]]

    def bbox(node, leaf):
        if leaf:
            return (node[0][0], node[0][1], node[-1][2], max(x[3] for x in node if x[2] == node[-1][2]))
        boxes = [bbox(ch, lf) for ch, lf in node]
        return (boxes[0][0], boxes[0][1], boxes[-1][2], max(b[3] for b in boxes if b[2] == boxes[-1][2]))

    # upper levels hold (child node, child_is_leaf)
    cur = [(nd, True) for nd in levels[0]]
    tree = [cur]
    while len(cur) > 1:
        cur = [(cur[i:i + _BLOCK], False) for i in range(0, len(cur), _BLOCK)]
        tree.append(cur)
    tree.reverse()  # root level first
    # layout
    pos = index_offset + len(head)
    offs = []
    for lv in tree:
        offs.append([])
        for node, leaf in lv:
            offs[-1].append(pos)
            pos += 4 + len(node) * (32 if leaf else 24)
    body = bytearray()
    for li, lv in enumerate(tree):
        child = 0
        for node, leaf in lv:
            body += struct.pack("<BBH", 1 if leaf else 0, 0, len(node))
            if leaf:
                for it in node:
                    body += struct.pack("<IIIIQQ", *it)
            else:
                for ch, ch_leaf in node:
                    bb = bbox(ch, ch_leaf)
                    body += struct.pack("<IIIIQ", bb[0], bb[1], bb[2], bb[3], offs[li + 1][child])
                    child += 1
    return head + bytes(body)


def write_fixed_step_bigwig(output_file, header, interval_scores) -> None:
    """``header``: list of ``(contig, length)``; ``interval_scores``: iterable of
    ``(contig, start, values)`` in header order.  An out-of-order or overlapping
    interval is skipped with a note on stderr (pyBigWig raises RuntimeError there
    and the reference skips the interval, frag/_multi_wps.py:319-325)."""
    def runs():
        for contig, start, values in interval_scores:
            values = np.asarray(values)
            yield contig, [int(start)], values, np.array([0, len(values)], np.int64)
    write_fixed_step_bigwig_runs(output_file, header, runs())


class RunOrder:
    """pyBigWig's ordering rule for ``addEntries`` as the reference meets it (frag/_multi_wps.py:319-325,
    frag/_cleavage_profile.py:470-486): an interval on an unknown contig, or one that starts before the end of
    the last accepted one, raises there and is skipped with a note on stderr.  The decision needs only the
    intervals' coordinates, so every rank of a multi-GPU run can take it before anything is computed."""

    def __init__(self, header, quiet: bool = False):
        self.chrom_id = {c: i for i, (c, _) in enumerate(header)}
        self.last = (-1, -1)
        self.quiet = quiet

    def keep(self, contig, starts, lengths) -> list:
        """Indices of the run's intervals that are written (empty intervals are dropped silently)."""
        cid = self.chrom_id.get(contig)
        kept = []
        for k, (st, n_k) in enumerate(zip(starts, lengths)):
            st, n_k = int(st), int(n_k)
            if n_k <= 0:  # (stop < start: a site whose midpoint lies beyond the contig end - as empty as stop == start)
                continue
            if cid is None or (cid, st) < self.last:
                if not self.quiet:
                    sys.stderr.write(f"{contig}:{st}-{st + n_k}\n invalid or out of order interval "
                                     "encountered. Skipping to next.\n")
                continue
            kept.append(k)
            self.last = (cid, st + n_k)
        return kept


def fixed_step_payload(cid: int, starts, values, offsets, threads: int = 0):
    """Data sections of one run of intervals on chromosome ``cid``: ``(blob, table, stats)`` of
    ``writers.bigwig_sections`` -- everything the container needs from the run, and what a rank that does not
    own the output file ships to the one that does."""
    from . import writers
    return writers.bigwig_sections(cid, starts, values, offsets, _ITEMS_PER_SECTION, 6, threads)


class FixedStepBigWigWriter:
    """The container around the runs' data sections: header, chromosome tree, R-tree index, one zoom level,
    total summary.  ``add`` takes the payloads in file order; ``close`` writes the indexes and patches the
    header."""

    def __init__(self, output_file, header):
        self.header = header
        tree, reloc = _chrom_tree(header)
        self.n_zoom = 1
        # the layout libBigWig (pyBigWig) writes: 64-byte header, room for ten zoom headers, the total summary, the
        # chromosome tree, the data (tests/test_bigwig.py holds a file of this writer against tests/data/test.bw,
        # which pyBigWig wrote, field by field)
        self.total_summary_off = 64 + 24 * _ZOOM_SLOTS
        self.chrom_tree_off = self.total_summary_off + 40
        self.data_off = self.chrom_tree_off + len(tree)
        tree = bytearray(tree)
        for r in reloc:
            (v,) = struct.unpack_from("<Q", tree, r)
            struct.pack_into("<Q", tree, r, v + self.chrom_tree_off)
        self.tree = tree
        self.leaf_items = []
        self.zoom_parts = []  # per run: (cid, table, stats)
        self.n_valid, self.vmin, self.vmax, self.vsum, self.vsq, self.max_raw = 0, np.inf, -np.inf, 0.0, 0.0, 0
        self.pos = self.data_off + 8
        self.fh = open(output_file, "wb")
        self.fh.write(b"\0" * self.pos)  # header, zoom header, chromosome tree, summary, section count: patched at the end

    def add(self, cid, blob, table, stats) -> None:
        if len(table) == 0:
            return
        self.fh.write(blob)
        sizes = table[:, 2]
        offs = self.pos + np.concatenate([[0], np.cumsum(sizes)[:-1]])
        self.leaf_items.extend(zip([cid] * len(table), table[:, 0].tolist(), [cid] * len(table), table[:, 1].tolist(),
                                   offs.tolist(), sizes.tolist()))
        self.pos += int(sizes.sum())
        self.zoom_parts.append((cid, table, stats))
        counts = table[:, 1] - table[:, 0]
        self.n_valid += int(counts.sum())
        self.max_raw = max(self.max_raw, 24 + 4 * int(counts.max()))
        self.vmin, self.vmax = min(self.vmin, float(stats[:, 0].min())), max(self.vmax, float(stats[:, 1].max()))
        self.vsum += float(stats[:, 2].sum())
        self.vsq += float(stats[:, 3].sum())

    def close(self) -> None:
        fh = self.fh
        vmin, vmax = self.vmin, self.vmax
        if self.n_valid == 0:
            vmin = vmax = 0.0
        index_off = self.pos
        index = _rtree(self.leaf_items, index_off)
        fh.write(index)
        # zoom level: one summary record per data section, in blocks of 512 records
        zoom_data_off = index_off + len(index)
        rec_dt = np.dtype([("cid", "<u4"), ("s", "<u4"), ("e", "<u4"), ("n", "<u4"), ("mn", "<f4"), ("mx", "<f4"),
                           ("sum", "<f4"), ("sq", "<f4")])
        recs = np.zeros(sum(len(t) for _, t, _ in self.zoom_parts), rec_dt)
        o = 0
        for cid, table, stats in self.zoom_parts:
            r = recs[o:o + len(table)]
            r["cid"], r["s"], r["e"], r["n"] = cid, table[:, 0], table[:, 1], table[:, 1] - table[:, 0]
            r["mn"], r["mx"], r["sum"], r["sq"] = stats[:, 0], stats[:, 1], stats[:, 2], stats[:, 3]
            o += len(table)
        zdata = bytearray(struct.pack("<I", len(recs)))
        zleaf = []
        zpos = zoom_data_off + 4
        max_raw = self.max_raw
        for o in range(0, len(recs), 512):
            blk = recs[o:o + 512]
            raw = blk.tobytes()
            comp = zlib.compress(raw, 6)
            zdata += comp
            zleaf.append((int(blk["cid"][0]), int(blk["s"][0]), int(blk["cid"][-1]), int(blk["e"][-1]), zpos, len(comp)))
            zpos += len(comp)
            max_raw = max(max_raw, len(raw))
        zoom_index_off = zpos
        zindex = _rtree(zleaf, zoom_index_off)
        fh.write(bytes(zdata))
        fh.write(zindex)
        fh.write(struct.pack("<I", _BW_MAGIC))
        fh.seek(0)
        fh.write(struct.pack("<IHHQQQHHQQIQ", _BW_MAGIC, 4, self.n_zoom, self.chrom_tree_off, self.data_off, index_off,
                             0, 0, 0, self.total_summary_off, max(max_raw, 1), 0))
        fh.write(struct.pack("<IIQQ", _ITEMS_PER_SECTION, 0, zoom_data_off, zoom_index_off))
        fh.write(b"\0" * (24 * (_ZOOM_SLOTS - self.n_zoom)))
        fh.write(struct.pack("<Qdddd", self.n_valid, vmin, vmax, self.vsum, self.vsq))
        fh.write(bytes(self.tree))
        fh.write(struct.pack("<Q", len(self.leaf_items)))
        fh.close()
        self.fh = None

    def __enter__(self):
        return self

    def __exit__(self, exc_type, *exc):
        if self.fh is not None:
            if exc_type is None:
                self.close()
            else:
                self.fh.close()


def select_intervals(keep, starts, values, offsets):
    """The kept intervals of a run, laid end to end again: ``(starts, values, offsets)``."""
    starts = np.asarray(starts, dtype=np.int64)
    offsets = np.asarray(offsets, dtype=np.int64)
    values = np.asarray(values)
    if len(keep) == len(starts):
        return starts, values, offsets
    vals = np.concatenate([values[offsets[k]:offsets[k + 1]] for k in keep]) if keep else values[:0]
    lens = np.array([offsets[k + 1] - offsets[k] for k in keep], np.int64)
    return starts[keep], vals, np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)


def write_fixed_step_bigwig_runs(output_file, header, contig_runs, threads: int = 0) -> None:
    """The same from runs of intervals: ``contig_runs`` yields ``(contig, starts, values, offsets)`` with interval
    ``k`` = ``values[offsets[k]:offsets[k+1]]`` at ``starts[k]`` (what one ``ftk_wps_intervals`` launch returns).
    The data sections -- float32 conversion, section headers, zlib, per-section summaries -- are built by
    the library's host threads (``writers.bigwig_sections``) and streamed to the file; ``FixedStepBigWigWriter``
    keeps the container."""
    order = RunOrder(header)
    with FixedStepBigWigWriter(output_file, header) as bw:
        for contig, starts, values, offsets in contig_runs:
            offsets = np.asarray(offsets, dtype=np.int64)
            keep = order.keep(contig, starts, np.diff(offsets))
            if not keep:
                continue
            starts, values, offsets = select_intervals(keep, starts, values, offsets)
            bw.add(order.chrom_id[contig], *fixed_step_payload(order.chrom_id[contig], starts, values, offsets, threads))


# ---------------------------------------------------------------------------
# reader
# ---------------------------------------------------------------------------
_BEDGRAPH_DT = np.dtype([("s", "<u4"), ("e", "<u4"), ("v", "<f4")])
_VARSTEP_DT = np.dtype([("s", "<u4"), ("v", "<f4")])


class BigWigFile:
    """Read side of pyBigWig as the reference uses it (frag/_adjust_wps.py:79-101): ``chroms()`` and
    ``intervals(contig, start, stop)``; data sections (bedGraph, varStep, fixedStep) are decoded with
    numpy, the most recently used ones are kept."""

    def __init__(self, path):
        self.path = str(path)
        b = self._b = open(path, "rb").read()
        if len(b) < 64:
            raise ValueError(f"{path} is not a bigWig file")
        magic, ver, nzoom, ct_off, data_off, idx_off, _, _, _, ts_off, ubuf, _ = struct.unpack_from("<IHHQQQHHQQIQ", b, 0)
        if magic != _BW_MAGIC:
            raise ValueError(f"{path} is not a little-endian bigWig file")
        self._compressed = bool(ubuf)
        tmagic, bsize, key, val, count, _ = struct.unpack_from("<IIIIQQ", b, ct_off)
        if tmagic != _CHROM_TREE_MAGIC:
            raise ValueError("bad chromosome tree")
        self._chroms = {}

        def walk_ct(off):
            leaf, _, n = struct.unpack_from("<BBH", b, off)
            off += 4
            for _ in range(n):
                k = b[off:off + key].rstrip(b"\0").decode()
                if leaf:
                    cid, size = struct.unpack_from("<II", b, off + key)
                    self._chroms[k] = (cid, size)
                else:
                    (child,) = struct.unpack_from("<Q", b, off + key)
                    walk_ct(child)
                off += key + 8

        if count:
            walk_ct(ct_off + 32)
        self._names = {cid: name for name, (cid, _) in self._chroms.items()}
        if struct.unpack_from("<I", b, idx_off)[0] != _RTREE_MAGIC:
            raise ValueError("bad R-tree index")
        leaves = []

        def walk_rt(off):
            leaf, _, n = struct.unpack_from("<BBH", b, off)
            off += 4
            for _ in range(n):
                if leaf:
                    leaves.append(struct.unpack_from("<IIIIQQ", b, off))
                    off += 32
                else:
                    (child,) = struct.unpack_from("<Q", b, off + 16)
                    walk_rt(child)
                    off += 24

        (n_blocks,) = struct.unpack_from("<Q", b, idx_off + 8)
        if n_blocks:
            walk_rt(idx_off + 48)
        self._leaves = np.array(leaves, dtype=np.int64).reshape(-1, 6)
        self._cache = {}

    def close(self):
        self._cache.clear()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def chroms(self):
        return {k: size for k, (_, size) in self._chroms.items()}

    def _section(self, k):
        sec = self._cache.get(k)
        if sec is None:
            doff, dsize = int(self._leaves[k, 4]), int(self._leaves[k, 5])
            raw = self._b[doff:doff + dsize]
            if self._compressed:
                raw = zlib.decompress(raw)
            cid, s0, e0, step, span, typ, _, n = struct.unpack_from("<IIIIIBBH", raw, 0)
            if typ == 1:
                rec = np.frombuffer(raw, _BEDGRAPH_DT, n, 24)
                st, en, v = rec["s"].astype(np.int64), rec["e"].astype(np.int64), rec["v"]
            elif typ == 2:
                rec = np.frombuffer(raw, _VARSTEP_DT, n, 24)
                st, v = rec["s"].astype(np.int64), rec["v"]
                en = st + span
            else:
                v = np.frombuffer(raw, "<f4", n, 24)
                st = s0 + np.arange(n, dtype=np.int64) * step
                en = st + span
            sec = (cid, st, en, v.astype(np.float64))
            if len(self._cache) >= 64:
                self._cache.pop(next(iter(self._cache)))
            self._cache[k] = sec
        return sec

    def sections(self):
        """All data sections in index order: ``(contig, starts, ends, values)``."""
        for k in range(len(self._leaves)):
            cid, st, en, v = self._section(k)
            yield self._names[cid], st, en, v

    def intervals(self, contig, start=0, stop=0):
        """Entries overlapping ``[start, stop)`` as ``(starts, ends, values)`` arrays, or ``None`` when
        there are none; RuntimeError for an unknown contig or bounds outside it (as pyBigWig)."""
        if contig not in self._chroms:
            raise RuntimeError("Invalid interval bounds!")
        cid, size = self._chroms[contig]
        if start == 0 and stop == 0:
            stop = size
        if start < 0 or stop > size or start >= stop:
            raise RuntimeError("Invalid interval bounds!")
        lv = self._leaves
        if len(lv) == 0:
            return None
        after_start = (lv[:, 2] > cid) | ((lv[:, 2] == cid) & (lv[:, 3] > start))
        before_stop = (lv[:, 0] < cid) | ((lv[:, 0] == cid) & (lv[:, 1] < stop))
        parts = []
        for k in np.flatnonzero(after_start & before_stop):
            scid, st, en, v = self._section(int(k))
            if scid != cid:
                continue
            keep = (en > start) & (st < stop)
            if keep.any():
                parts.append((st[keep], en[keep], v[keep]))
        if not parts:
            return None
        return tuple(np.concatenate([p[i] for p in parts]) for i in range(3))


    def values(self, contig, start, stop):
        """Per-base values of ``[start, stop)`` as float64, NaN where the file has no entry; RuntimeError for
        an unknown contig or bounds outside it (pyBigWig's ``values``, utils/_agg_bw.py:87)."""
        got = self.intervals(contig, int(start), int(stop))
        out = np.full(int(stop) - int(start), np.nan)
        if got is not None:
            st, en, v = got
            s0 = np.maximum(st, start) - start
            lens = np.minimum(en, stop) - start - s0
            if np.all(lens == 1):  # per-base tracks (what the WPS writers produce)
                out[s0] = v
            else:
                first = np.cumsum(lens) - lens
                out[np.repeat(s0 - first, lens) + np.arange(int(lens.sum()))] = np.repeat(v, lens)
        return out


_ZOOM_DT = np.dtype([("cid", "<u4"), ("s", "<u4"), ("e", "<u4"), ("n", "<u4"), ("mn", "<f4"), ("mx", "<f4"),
                     ("sum", "<f4"), ("sq", "<f4")])
_HEADER_FIELDS = ("magic", "version", "zoomLevels", "chromTreeOffset", "fullDataOffset", "fullIndexOffset", "fieldCount",
                  "definedFieldCount", "autoSqlOffset", "totalSummaryOffset", "uncompressBufSize", "extensionOffset")
_RTREE_FIELDS = ("magic", "blockSize", "itemCount", "startChromIx", "startBase", "endChromIx", "endBase", "endFileOffset",
                 "itemsPerSlot", "reserved")


def _rtree_leaves(b, off):
    """Leaf items ``(chromIx, start, chromIx, end, offset, size)`` of the R-tree at ``off``, in tree order."""
    head = dict(zip(_RTREE_FIELDS, struct.unpack_from("<IIQIIIIQII", b, off)))
    leaves = []

    def walk(at):
        leaf, _, n = struct.unpack_from("<BBH", b, at)
        at += 4
        for _ in range(n):
            if leaf:
                leaves.append(struct.unpack_from("<IIIIQQ", b, at))
                at += 32
            else:
                walk(struct.unpack_from("<Q", b, at + 16)[0])
                at += 24

    if head["itemCount"]:
        walk(off + 48)
    return head, leaves


def describe(path) -> dict:
    """Every field of a bigWig container a consumer (pyBigWig / libBigWig, IGV, bigWigToWig) reads, as plain data:
    the 64-byte header, the zoom headers, the chromosome B+ tree (header, items), the total summary, the R-tree
    header and leaves, every data section's 24-byte header and values (inflated), every zoom level's records and
    index.  What tests compare between a file of this package's writer and one pyBigWig wrote."""
    b = open(path, "rb").read()
    hdr = dict(zip(_HEADER_FIELDS, struct.unpack_from("<IHHQQQHHQQIQ", b, 0)))
    out = dict(header=hdr, file_bytes=len(b), trailer_magic=struct.unpack_from("<I", b, len(b) - 4)[0])
    ct = hdr["chromTreeOffset"]
    tree = dict(zip(("magic", "blockSize", "keySize", "valSize", "itemCount", "reserved"), struct.unpack_from("<IIIIQQ", b, ct)))
    items = []

    def walk_ct(at):
        leaf, _, n = struct.unpack_from("<BBH", b, at)
        at += 4
        for _ in range(n):
            if leaf:
                items.append((b[at:at + tree["keySize"]], *struct.unpack_from("<II", b, at + tree["keySize"])))
            else:
                walk_ct(struct.unpack_from("<Q", b, at + tree["keySize"])[0])
            at += tree["keySize"] + 8

    if tree["itemCount"]:
        walk_ct(ct + 32)
    tree["items"] = items
    out["chrom_tree"] = tree
    out["total_summary"] = dict(zip(("validCount", "minVal", "maxVal", "sumData", "sumSquares"),
                                    struct.unpack_from("<Qdddd", b, hdr["totalSummaryOffset"]))) if hdr["totalSummaryOffset"] else None
    out["section_count"] = struct.unpack_from("<Q", b, hdr["fullDataOffset"])[0]
    rhead, leaves = _rtree_leaves(b, hdr["fullIndexOffset"])
    out["rtree"] = dict(rhead, leaves=leaves)
    sections = []
    for leaf in leaves:
        raw = b[leaf[4]:leaf[4] + leaf[5]]
        if hdr["uncompressBufSize"]:
            raw = zlib.decompress(raw)
        sh = dict(zip(("chromId", "chromStart", "chromEnd", "itemStep", "itemSpan", "type", "reserved", "itemCount"),
                      struct.unpack_from("<IIIIIBBH", raw, 0)))
        sh["raw_bytes"] = len(raw)
        sh["payload"] = raw[24:]
        sections.append(sh)
    out["sections"] = sections
    zooms = []
    for k in range(hdr["zoomLevels"]):
        red, rsv, zd, zi = struct.unpack_from("<IIQQ", b, 64 + 24 * k)
        zhead, zleaves = _rtree_leaves(b, zi)
        recs = []
        for leaf in zleaves:
            raw = b[leaf[4]:leaf[4] + leaf[5]]
            if hdr["uncompressBufSize"]:
                raw = zlib.decompress(raw)
            recs.append(np.frombuffer(raw, _ZOOM_DT))
        zooms.append(dict(reductionLevel=red, reserved=rsv, dataOffset=zd, indexOffset=zi,
                          recordCount=struct.unpack_from("<I", b, zd)[0], rtree=dict(zhead, leaves=zleaves),
                          records=np.concatenate(recs) if recs else np.zeros(0, _ZOOM_DT)))
    out["zoom"] = zooms
    return out


def verify(path, strict: bool = True) -> list:
    """Check a bigWig container against its own data - the parts ``BigWigFile`` does not need to answer queries but
    other readers do: offsets in range, the trailer magic, section count = R-tree item count, every R-tree leaf's
    bounds = its section's header, the R-tree header's bounds = the leaves' extremes, ``uncompressBufSize`` >= every
    inflated section, the total summary (validCount / min / max / sum / sumSquares) recomputed from the sections, every
    zoom record (bases covered, min, max, sum, sum of squares over the data inside its range) and the zoom index's
    bounds.  Raises ``ValueError`` on the first violation.  ``strict=False`` tolerates the two things libBigWig itself
    writes differently (a fixedStep section's chromEnd computed from the buffer length INCLUDING its 24-byte header,
    i.e. 6 steps too far; zoom records whose sums are left zero) and returns them as notes."""
    d = describe(path)
    notes = []

    def bad(msg):
        raise ValueError(f"{path}: {msg}")

    def quirk(msg):
        if strict:
            bad(msg)
        notes.append(msg)

    h = d["header"]
    if h["magic"] != _BW_MAGIC or d["trailer_magic"] != _BW_MAGIC:
        bad("magic numbers")
    for k in ("chromTreeOffset", "fullDataOffset", "fullIndexOffset", "totalSummaryOffset"):
        if not 64 <= h[k] < d["file_bytes"]:
            bad(f"{k} {h[k]} outside the file")
    if not 1 <= h["zoomLevels"] <= _ZOOM_SLOTS:
        bad(f"{h['zoomLevels']} zoom levels")
    if d["chrom_tree"]["magic"] != _CHROM_TREE_MAGIC or d["chrom_tree"]["valSize"] != 8 or \
            d["chrom_tree"]["itemCount"] != len(d["chrom_tree"]["items"]):
        bad("chromosome tree")
    keys = [it[0] for it in d["chrom_tree"]["items"]]
    if keys != sorted(keys) or any(len(k) != d["chrom_tree"]["keySize"] for k in keys):
        bad("chromosome tree keys must be sorted and keySize wide")
    sizes = {it[1]: it[2] for it in d["chrom_tree"]["items"]}
    r = d["rtree"]
    if r["magic"] != _RTREE_MAGIC or r["itemCount"] != len(r["leaves"]) or d["section_count"] != len(r["leaves"]):
        bad("R-tree item count / section count")
    covered = {}  # chromId -> list of (start, end, value) runs, as arrays
    n_valid, vmin, vmax, vsum, vsq = 0, np.inf, -np.inf, 0.0, 0.0
    for leaf, sec in zip(r["leaves"], d["sections"]):
        if sec["raw_bytes"] > h["uncompressBufSize"]:
            bad(f"section of {sec['raw_bytes']} bytes exceeds uncompressBufSize {h['uncompressBufSize']}")
        n = sec["itemCount"]
        if sec["type"] == 3:
            v = np.frombuffer(sec["payload"], "<f4", n)
            st = sec["chromStart"] + np.arange(n, dtype=np.int64) * sec["itemStep"]
            en = st + sec["itemSpan"]
        elif sec["type"] == 2:
            rec = np.frombuffer(sec["payload"], _VARSTEP_DT, n)
            st, v = rec["s"].astype(np.int64), rec["v"]
            en = st + sec["itemSpan"]
        else:
            rec = np.frombuffer(sec["payload"], _BEDGRAPH_DT, n)
            st, en, v = rec["s"].astype(np.int64), rec["e"].astype(np.int64), rec["v"]
        true_end = int(en.max()) if n else sec["chromStart"]
        if sec["chromId"] not in sizes or true_end > sizes[sec["chromId"]]:
            bad("section outside its chromosome")
        if sec["chromEnd"] != true_end:
            quirk(f"section chromEnd {sec['chromEnd']} but its last item ends at {true_end}")
        if (leaf[0], leaf[1], leaf[2], leaf[3]) != (sec["chromId"], sec["chromStart"], sec["chromId"], sec["chromEnd"]):
            bad(f"R-tree leaf {leaf[:4]} does not match its section header")
        covered.setdefault(sec["chromId"], []).append((st, en, v.astype(np.float64)))
        w = (en - st).astype(np.float64)
        n_valid += int(w.sum())
        if n:
            vmin, vmax = min(vmin, float(v.min())), max(vmax, float(v.max()))
        vsum += float((v.astype(np.float64) * w).sum())
        vsq += float((v.astype(np.float64) ** 2 * w).sum())
    if r["leaves"]:
        first, last = r["leaves"][0], r["leaves"][-1]
        if (r["startChromIx"], r["startBase"], r["endChromIx"], r["endBase"]) != (first[0], first[1], last[2], last[3]):
            bad("R-tree header bounds are not the leaves' extremes")
        if r["endFileOffset"] != last[4] + last[5]:
            quirk(f"R-tree endFileOffset {r['endFileOffset']} is not the end of the indexed data {last[4] + last[5]}")
    ts = d["total_summary"]
    if ts is None:
        bad("no total summary")
    want = (n_valid, vmin if n_valid else 0.0, vmax if n_valid else 0.0, vsum, vsq)
    got = (ts["validCount"], ts["minVal"], ts["maxVal"], ts["sumData"], ts["sumSquares"])
    if got[0] != want[0] or not np.allclose(got[1:], want[1:], rtol=1e-6, atol=1e-9):
        bad(f"total summary {got} but the data give {want}")
    for z in d["zoom"]:
        recs = z["records"]
        if z["recordCount"] != len(recs):
            bad("zoom record count")
        for rec in recs:
            runs = covered.get(int(rec["cid"]), [])
            n = 0
            mn, mx, sm, sq = np.inf, -np.inf, 0.0, 0.0
            for st, en, v in runs:
                lo, hi = np.maximum(st, int(rec["s"])), np.minimum(en, int(rec["e"]))
                w = np.maximum(hi - lo, 0).astype(np.float64)
                if w.any():
                    n += int(w.sum())
                    mn, mx = min(mn, float(v[w > 0].min())), max(mx, float(v[w > 0].max()))
                    sm += float((v * w).sum())
                    sq += float((v * v * w).sum())
            if int(rec["n"]) != n or (n and not np.allclose([rec["mn"], rec["mx"]], [mn, mx], rtol=1e-6, atol=1e-9)):
                bad(f"zoom record {rec} does not describe the data in its range (valid {n}, min {mn}, max {mx})")
            if n and not np.allclose([rec["sum"], rec["sq"]], [sm, sq], rtol=1e-5, atol=1e-6):
                if float(rec["sum"]) == 0.0 and float(rec["sq"]) == 0.0:
                    quirk(f"zoom record of {rec['cid']}:{rec['s']}-{rec['e']} carries zero sums (data: {sm}, {sq})")
                else:
                    bad(f"zoom record sums {rec['sum']}, {rec['sq']} but the data give {sm}, {sq}")
        zl = z["rtree"]["leaves"]
        if z["rtree"]["magic"] != _RTREE_MAGIC or z["rtree"]["itemCount"] != len(zl):
            bad("zoom index")
        if len(recs) and zl:
            if (z["rtree"]["startChromIx"], z["rtree"]["startBase"]) != (int(recs[0]["cid"]), int(recs[0]["s"])) or \
                    (z["rtree"]["endChromIx"], z["rtree"]["endBase"]) != (int(recs[-1]["cid"]), int(recs[-1]["e"])):
                bad("zoom index bounds are not its records' extremes")
    return notes


def read_bigwig(path):
    """Return ``(chroms, intervals)``: ``chroms`` = {name: (id, size)};
    ``intervals`` = list of ``(chrom_name, start, end, value)`` decoded from
    every data section, in file order.  The container is checked on the way (``verify``: summary, index bounds,
    zoom records; libBigWig's own two deviations are tolerated)."""
    verify(path, strict=False)
    bw = BigWigFile(path)
    out = []
    for name, st, en, v in bw.sections():
        f32 = v.astype(np.float32)
        out.extend((name, int(a), int(b), float(c)) for a, b, c in zip(st, en, f32))
    return dict(bw._chroms), out
