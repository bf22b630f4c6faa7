"""
Seeded synthetic cfDNA fragments (the workload generator of BASELINE.md
section 4 / SURVEY.md section 8-d).  NumPy version: used by the parity tests
and small benches; bench.py has a device-side generator of the same
distribution for whole-genome sizes.
"""
from __future__ import annotations

import numpy as np

# b37 / hg19 primary contigs (tests/data/b37.chrom.sizes of the reference)
B37_SIZES = {
    "1": 249250621, "2": 243199373, "3": 198022430, "4": 191154276, "5": 180915260, "6": 171115067,
    "7": 159138663, "8": 146364022, "9": 141213431, "10": 135534747, "11": 135006516, "12": 133851895,
    "13": 115169878, "14": 107349540, "15": 102531392, "16": 90354753, "17": 81195210, "18": 78077248,
    "19": 59128983, "20": 63025520, "21": 48129895, "22": 51304566, "X": 155270560, "Y": 59373566,
}
SEED_BASE = 20260723


def n_fragments(contig_len: int, depth: float) -> int:
    """N_frag(contig) = round(depth * contig_len / 300)  (2 x 150 bp pairs)."""
    return int(round(depth * contig_len / 300.0))


def synth_contig(contig_len: int, depth: float = 30.0, seed: int = SEED_BASE, n: int | None = None):
    """Return start-sorted SoA columns (start i32, end i32, mapq u8, strand u8)."""
    rng = np.random.default_rng(seed)
    if n is None:
        n = n_fragments(contig_len, depth)
    start = rng.integers(0, max(contig_len - 1000, 1), size=n, dtype=np.int64)
    u = rng.random(n)
    length = np.where(
        u < 0.85, rng.normal(167.0, 12.0, n),
        np.where(u < 0.97, rng.normal(334.0, 25.0, n), rng.uniform(30.0, 600.0, n)))
    length = np.clip(np.rint(length), 30, 1000).astype(np.int64)
    end = start + length
    mapq = np.where(rng.random(n) < 0.85, 60, rng.integers(0, 60, size=n)).astype(np.uint8)
    strand = (rng.random(n) < 0.5).astype(np.uint8)
    order = np.lexsort((end, start))
    return (start[order].astype(np.int32), end[order].astype(np.int32), mapq[order], strand[order])


def tiling_windows(contig_len: int, width: int):
    """Non-overlapping windows [k*width, min((k+1)*width, contig_len))."""
    ws = np.arange(0, contig_len, width, dtype=np.int64)
    we = np.minimum(ws + width, contig_len)
    return ws.astype(np.int32), we.astype(np.int32)


# ---- whole-genome workload pieces (bench.py, tools/, the full-size GPU tests) ---------------------------------
_WINDOW = 100_000


def gen_contig_device(torch, dev, contig_len, n, seed):
    """Seeded device-side generator of the BASELINE.md mixture; start-sorted SoA tensors."""
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    start = torch.randint(0, max(contig_len - 1000, 1), (n,), generator=g, device=dev, dtype=torch.int64)
    u = torch.rand(n, generator=g, device=dev)
    z = torch.randn(n, generator=g, device=dev)
    v = torch.rand(n, generator=g, device=dev)
    length = torch.where(u < 0.85, 167.0 + 12.0 * z, torch.where(u < 0.97, 334.0 + 25.0 * z, 30.0 + 570.0 * v))
    length = torch.clamp(torch.round(length), 30, 1000).to(torch.int64)
    del u, z, v
    key, _ = torch.sort(start * 2048 + length)  # sort by (start, end)
    del start, length
    s = (key >> 11).to(torch.int32)
    e = (s + (key & 2047).to(torch.int32)).contiguous()
    del key
    m = torch.rand(n, generator=g, device=dev)
    mapq = torch.where(m < 0.85, torch.full((n,), 60, device=dev, dtype=torch.int64),
                       torch.randint(0, 60, (n,), generator=g, device=dev)).to(torch.uint8)
    strand = (torch.rand(n, generator=g, device=dev) < 0.5).to(torch.uint8)
    return s, e, mapq, strand


def synth_gaps(contig_len):
    """Synthetic centromere / telomere constants (hg19-like proportions)."""
    c0 = int(contig_len * 0.40) // _WINDOW * _WINDOW
    return (c0, c0 + 3_000_000, [(0, 10_000), (contig_len - 10_000, contig_len)])


def synth_blacklist(contig_len, seed, n_regions):
    rng = np.random.default_rng(seed)
    s = np.sort(rng.integers(0, contig_len - 6000, n_regions)).astype(np.int32)
    e = (s + rng.integers(200, 5000, n_regions)).astype(np.int32)
    order = np.lexsort((e, s))
    return s[order], e[order]


def write_paired_bam(path, contig, size, depth, seed, read_len=50):
    """Coordinate-sorted paired-end BAM, one pair per synthetic fragment (numpy-built fixed-size records,
    BGZF blocks deflated by the library's writer).  Returns the decoder's expected columns: fragments in
    start order with their read1 span."""
    import os
    import struct
    from . import bgzf, writers
    s, e, q, st = synth_contig(size, depth, seed)
    e = np.maximum(e, s + read_len).astype(np.int32)
    n = len(s)
    ln = (e - s).astype(np.int64)
    fwd = st == 1
    r1_pos = np.where(fwd, s, e - read_len).astype(np.int64)
    r2_pos = np.where(fwd, e - read_len, s).astype(np.int64)
    rec = np.dtype([("block_size", "<i4"), ("ref", "<i4"), ("pos", "<i4"), ("l_name", "u1"), ("mapq", "u1"),
                    ("bin", "<u2"), ("n_cigar", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("next_ref", "<i4"),
                    ("next_pos", "<i4"), ("tlen", "<i4"), ("name", "S8"), ("cigar", "<u4"),
                    ("seq", "u1", (read_len // 2,)), ("qual", "u1", (read_len,))])
    a = np.zeros(2 * n, rec)
    a["block_size"] = rec.itemsize - 4
    a["l_name"], a["n_cigar"], a["l_seq"], a["cigar"] = 8, 1, read_len, read_len << 4
    a["pos"][:n], a["pos"][n:] = r1_pos, r2_pos
    a["next_pos"][:n], a["next_pos"][n:] = r2_pos, r1_pos
    a["mapq"][:n] = a["mapq"][n:] = q
    a["tlen"][:n], a["tlen"][n:] = np.where(fwd, ln, -ln), np.where(fwd, -ln, ln)
    a["flag"][:n], a["flag"][n:] = np.where(fwd, 99, 83), np.where(fwd, 147, 163)
    digits = np.zeros((n, 8), np.uint8)  # the pair's number as seven digits and the NUL that l_read_name counts
    digits[:, :7] = np.arange(n, dtype=np.int64)[:, None] // 10 ** np.arange(6, -1, -1, dtype=np.int64)[None, :] % 10 + 48
    a["name"][:n] = a["name"][n:] = digits.view("S8")[:, 0]
    rng = np.random.default_rng(seed)
    a["seq"] = rng.integers(0, 256, (2 * n, read_len // 2), dtype=np.uint8)
    lut = np.repeat(np.array([2, 11, 25, 37], np.uint8), [8, 18, 51, 179])  # 3 / 7 / 20 / 70 % of the qualities
    a["qual"] = lut[rng.integers(0, 256, (2 * n, read_len), dtype=np.uint8)]
    order = np.argsort(a["pos"], kind="stable")
    a = a[order]
    text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:%s\tLN:%d\n" % (contig.encode(), size)
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1)
    head += struct.pack("<i", len(contig) + 1) + contig.encode() + b"\0" + struct.pack("<i", size)
    offs = writers.bgzf_write(path, head, level=1, write_eof=False)
    offs2 = writers.bgzf_write(path, a.tobytes(), level=1, append=True)
    # (with the 16 kb linear index: a reader may start inside the contig - ftk_fragstream_open_region)
    linear = bgzf.linear_index(a["pos"], a["pos"].astype(np.int64) + read_len, np.full(2 * n, rec.itemsize, np.int64), offs2)
    bgzf.write_index(str(path) + ".bai", True, [(contig, int(offs2[0]) << 16, int(offs2[-1]) << 16)], [linear])
    # read1 records in file order -> stable sort by fragment start = the decoder's row order
    is_r1 = order < n
    file_rank = order[is_r1]                          # fragment index of every read1 record, in file order
    by_start = np.argsort(s[file_rank], kind="stable")
    rows = file_rank[by_start]
    return dict(s=s[rows], e=e[rows], q=q[rows], st=st[rows], r1s=r1_pos[rows].astype(np.int32),
                r1e=(r1_pos[rows] + read_len).astype(np.int32), n=n, file_bytes=os.path.getsize(path))


def write_paired_bam_contigs(path, contigs, depth, seed, read_len=50, step=8_388_608):
    """Coordinate-sorted paired-end BAM of SEVERAL contigs at any size (BASELINE config 5 at real scale: a chr1-sized
    60x contig is 5.7 GB on disk): records are built and written in position windows of ``step`` bases, so host memory
    stays bounded by one window's records whatever the file size.  ``contigs``: ``[(name, length)]`` in header order;
    fragments of contig ``k`` are ``synth_contig(length, depth, seed + k)`` (ends stretched to ``read_len``).  The
    ``.bai`` carries every contig's span and its 16 kb linear index, so a reader may start inside a contig
    (``ftk_fragstream_open_region``).  Returns ``{name: dict(s, e, q, st, r1s, r1e, n, first_off, end_off)}``: the
    fragments in start order with their read1 span, and the file offsets of the contig's first block / the end of its
    last one (which, with the 16 kb ``linear`` index, say whether a region read lies behind the 4 GiB mark)."""
    import os
    import struct
    from . import bgzf, writers
    name_len = 10
    rec = np.dtype([("block_size", "<i4"), ("ref", "<i4"), ("pos", "<i4"), ("l_name", "u1"), ("mapq", "u1"),
                    ("bin", "<u2"), ("n_cigar", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("next_ref", "<i4"),
                    ("next_pos", "<i4"), ("tlen", "<i4"), ("name", f"S{name_len}"), ("cigar", "<u4"),
                    ("seq", "u1", (read_len // 2,)), ("qual", "u1", (read_len,))])
    text = b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % (c.encode(), n) for c, n in contigs)
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(contigs))
    for c, n in contigs:
        head += struct.pack("<i", len(c) + 1) + c.encode() + b"\0" + struct.pack("<i", n)
    writers.bgzf_write(path, head, level=1, write_eof=False)
    lut = np.repeat(np.array([2, 11, 25, 37], np.uint8), [8, 18, 51, 179])  # 3 / 7 / 20 / 70 % of the qualities
    rng = np.random.default_rng(seed)
    powers = 10 ** np.arange(name_len - 2, -1, -1, dtype=np.int64)
    out, spans, linear = {}, [], []
    for k, (c, size) in enumerate(contigs):
        s, e, q, st = synth_contig(size, depth, seed + k)
        e = np.maximum(e, s + read_len).astype(np.int32)
        fwd = st == 1
        ln = (e - s).astype(np.int64)
        r1_pos = np.where(fwd, s, e - read_len).astype(np.int64)
        r2_pos = np.where(fwd, e - read_len, s).astype(np.int64)
        never = np.uint64(0xFFFFFFFFFFFFFFFF)
        lin = np.full(((size + read_len) >> 14) + 1, never, np.uint64)
        first_off = end_off = None
        n_records = 0
        for a in range(0, size, step):
            b = min(a + step, size)
            lo, hi = int(np.searchsorted(s, a - 1000)), int(np.searchsorted(s, b))  # (fragments are <= 1000 long)
            idx = np.arange(lo, hi)
            m = len(idx)
            blk = np.zeros(2 * m, rec)
            blk["block_size"] = rec.itemsize - 4
            blk["ref"] = blk["next_ref"] = k
            blk["l_name"], blk["n_cigar"], blk["l_seq"], blk["cigar"] = name_len, 1, read_len, read_len << 4
            blk["pos"][:m], blk["pos"][m:] = r1_pos[idx], r2_pos[idx]
            blk["next_pos"][:m], blk["next_pos"][m:] = r2_pos[idx], r1_pos[idx]
            blk["mapq"][:m] = blk["mapq"][m:] = q[idx]
            blk["tlen"][:m], blk["tlen"][m:] = np.where(fwd[idx], ln[idx], -ln[idx]), np.where(fwd[idx], -ln[idx], ln[idx])
            blk["flag"][:m], blk["flag"][m:] = np.where(fwd[idx], 99, 83), np.where(fwd[idx], 147, 163)
            digits = np.zeros((m, name_len), np.uint8)  # the pair's number in decimal + the NUL l_read_name counts
            digits[:, :name_len - 1] = idx[:, None] // powers[None, :] % 10 + 48
            blk["name"][:m] = blk["name"][m:] = digits.view(f"S{name_len}")[:, 0]
            blk = blk[(blk["pos"] >= a) & (blk["pos"] < b)]
            blk["seq"] = rng.integers(0, 256, (len(blk), read_len // 2), dtype=np.uint8)
            blk["qual"] = lut[rng.integers(0, 256, (len(blk), read_len), dtype=np.uint8)]
            blk = blk[np.argsort(blk["pos"], kind="stable")]
            if not len(blk):
                continue
            n_records += len(blk)
            offs = writers.bgzf_write(path, blk.tobytes(), level=1, append=True, write_eof=False)
            if first_off is None:
                first_off = int(offs[0])
            end_off = int(offs[-1])
            # linear index: the first record (lowest virtual offset) overlapping every 16 kb window; a record covers at
            # most two windows, records of a window come in file order, windows only ever see later chunks afterwards
            at = np.arange(len(blk), dtype=np.int64) * rec.itemsize
            voff = (offs[at // 0xFF00].astype(np.uint64) << np.uint64(16)) | (at % 0xFF00).astype(np.uint64)
            pos = blk["pos"].astype(np.int64)
            for w in (pos >> 14, (pos + read_len - 1) >> 14):
                uw, first = np.unique(w, return_index=True)
                lin[uw] = np.minimum(lin[uw], voff[first])
            del blk, digits
        assert n_records == 2 * len(s)
        nxt = np.uint64(0)
        for w in range(len(lin) - 1, -1, -1):  # (htslib: a window without records points at the next one's)
            if lin[w] == never:
                lin[w] = nxt
            else:
                nxt = lin[w]
        spans.append((c, (first_off or 0) << 16, (end_off or 0) << 16))
        linear.append(lin)
        out[c] = dict(s=s, e=e, q=q, st=st, r1s=r1_pos.astype(np.int32), r1e=(r1_pos + read_len).astype(np.int32),
                      n=len(s), first_off=first_off, end_off=end_off, linear=lin)
    with open(path, "ab") as fh:
        fh.write(bgzf._EOF)
    bgzf.write_index(str(path) + ".bai", True, spans, linear)
    return out


def write_paired_bam_native(path, contigs, depth, seed, read_len=50, fragments=None, level=1, threads=0,
                            keep=("s", "e", "q", "st", "r1s", "r1e")):
    """The file ``write_paired_bam_contigs`` writes, at the rate of the host threads instead of numpy's: every contig's
    records are built, sorted and deflated in C (``ftk_synth_bam_contig``: 512 kb position windows, libdeflate), so a
    whole-genome 60x BAM (BASELINE config 5: 1.2 G records, ~145 GB of record bytes) is a matter of the disk.  Same record
    layout and order; sequence / quality bytes come from a counter-based generator instead of numpy's.
    ``fragments``: optional ``callable(k, name, size) -> (s, e, q, st)`` (e.g. columns generated on the GPU); default
    ``synth_contig(size, depth, seed + k)``.  Ends are stretched to ``read_len``.  Returns the same dict per contig
    (``keep`` names the columns worth holding on to: a whole genome's are 11 GB)."""
    import ctypes as C
    import struct
    from . import _lib as L, bgzf, writers
    lib = L.load()
    name_len = 10
    text = b"@HD\tVN:1.6\tSO:coordinate\n" + b"".join(b"@SQ\tSN:%s\tLN:%d\n" % (c.encode(), n) for c, n in contigs)
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", len(contigs))
    for c, n in contigs:
        head += struct.pack("<i", len(c) + 1) + c.encode() + b"\0" + struct.pack("<i", n)
    writers.bgzf_write(path, head, level=1, write_eof=False)
    out, spans, linear = {}, [], []
    never = np.uint64(0xFFFFFFFFFFFFFFFF)
    for k, (c, size) in enumerate(contigs):
        s, e, q, st = fragments(k, c, size) if fragments else synth_contig(size, depth, seed + k)
        s = np.ascontiguousarray(s, np.int32)
        e = np.maximum(e, s + read_len).astype(np.int32)
        q, st = np.ascontiguousarray(q, np.uint8), np.ascontiguousarray(st, np.uint8)
        lin = np.full(((size + read_len) >> 14) + 1, never, np.uint64)
        first, end, n_rec = C.c_int64(-1), C.c_int64(-1), C.c_int64(0)
        rc = lib.ftk_synth_bam_contig(str(path).encode(), k, int(size), L.ptr(s), L.ptr(e), L.ptr(q), L.ptr(st), len(s),
                                      int(read_len), name_len, int(seed) + k, int(level), int(threads), L.ptr(lin), len(lin),
                                      C.byref(first), C.byref(end), C.byref(n_rec))
        if rc != L.FTK_OK:
            raise OSError(lib.ftk_fragtable_error().decode())
        assert n_rec.value == 2 * len(s), (c, n_rec.value, len(s))
        has = lin != never  # (htslib: a window without records points at the next one's; behind the last: 0)
        idx = np.where(has, np.arange(len(lin)), len(lin))
        nxt = np.minimum.accumulate(idx[::-1])[::-1]
        lin = np.where(nxt < len(lin), lin[np.minimum(nxt, len(lin) - 1)], np.uint64(0))
        first_off = first.value if first.value >= 0 else None
        spans.append((c, (first_off or 0) << 16, (end.value if first_off is not None else 0) << 16))
        linear.append(lin)
        fwd = st == 1
        d = dict(n=len(s), first_off=first_off, end_off=end.value if first_off is not None else None, linear=lin)
        cols = dict(s=s, e=e, q=q, st=st)
        if "r1s" in keep or "r1e" in keep:
            r1 = np.where(fwd, s, e - read_len).astype(np.int32)
            cols["r1s"], cols["r1e"] = r1, (r1 + read_len).astype(np.int32)
        d.update({name: v for name, v in cols.items() if name in keep})
        out[c] = d
    with open(path, "ab") as fh:
        fh.write(bgzf._EOF)
    bgzf.write_index(str(path) + ".bai", True, spans, linear)
    return out


# ---- BASELINE config 5: a whole-genome 60x BAM -------------------------------------------------------------------------
GENOME_BAM_SEED = 5150
GENOME_BAM_READ = 50


def genome_bam_contigs(scale: float = 1.0):
    """The 24 b37 contigs, each at ``scale`` of its length (1.0: the real genome - 6.2 x 10^8 pairs at 60x, ~70 GB of
    BAM; a smaller scale keeps every contig and shortens them all, for boxes that cannot hold the file)."""
    return [(c, max(int(n * scale), 200_000)) for c, n in B37_SIZES.items()]


def genome_bam_fragments(k: int, size: int, depth: float = 60.0, torch=None, dev=None):
    """Contig ``k``'s fragments of the whole-genome BAM (start-sorted host columns, ends stretched to the read length) -
    from the seeded generator on the device when torch is given (0.1 s for chr1), else numpy's.  The writer and the
    checks call this with the same arguments, so nothing of a 6 x 10^8-pair genome has to be kept in between."""
    n = n_fragments(size, depth)
    if torch is not None:
        s, e, q, st = (t.cpu().numpy() for t in gen_contig_device(torch, dev, size, n, GENOME_BAM_SEED + k))
    else:
        s, e, q, st = synth_contig(size, depth, GENOME_BAM_SEED + k)
    e = np.maximum(e, s + GENOME_BAM_READ).astype(np.int32)
    return s, e, q, st


def genome_bam_expected(k: int, size: int, depth: float = 60.0, torch=None, dev=None):
    """What the decoder must hand out for contig ``k``: the fragments with their read1 span (what the tests' checker compares with)."""
    s, e, q, st = genome_bam_fragments(k, size, depth, torch, dev)
    r1 = np.where(st == 1, s, e - GENOME_BAM_READ).astype(np.int32)
    return dict(s=s, e=e, q=q, st=st, r1s=r1, r1e=(r1 + GENOME_BAM_READ).astype(np.int32), n=len(s))


def write_genome_bam(path, scale: float = 1.0, depth: float = 60.0, torch=None, dev=None, threads: int = 0):
    """BASELINE config 5's input: ONE coordinate-sorted paired-end BAM of all 24 contigs (``genome_bam_contigs``) at
    ``depth``, with its ``.bai`` (spans + 16 kb linear index), written by ``write_paired_bam_native``.  Returns
    ``(contigs, info)``: ``info[name]`` = ``dict(n, first_off, end_off, linear)``."""
    contigs = genome_bam_contigs(scale)
    info = write_paired_bam_native(path, contigs, depth, GENOME_BAM_SEED, read_len=GENOME_BAM_READ,
                                   fragments=lambda k, c, size: genome_bam_fragments(k, size, depth, torch, dev),
                                   threads=threads, keep=())
    return contigs, info


_BAM_BYTES_PER_RECORD = 57.6  # measured: 2.302 GB for 2 x 20 M records of 125 bytes deflated at level 1


def genome_bam_bytes(scale: float = 1.0, depth: float = 60.0) -> int:
    """Expected size of the whole-genome BAM at ``scale`` (71 GB at 1.0 / 60x)."""
    return int(2 * sum(n_fragments(n, depth) for _, n in genome_bam_contigs(scale)) * _BAM_BYTES_PER_RECORD)


def big_scratch_dir(bytes_needed: int) -> str:
    """A directory that can hold ``bytes_needed`` with room to spare: ``FTK_BIG_TMP``, the temp directory, or - when that
    is too small and the machine has the memory - ``/dev/shm``; else whichever has the most free space."""
    import os
    import shutil
    import tempfile
    cands = [d for d in (os.environ.get("FTK_BIG_TMP"), tempfile.gettempdir(), "/dev/shm") if d and os.path.isdir(d)]

    def free(d):
        f = shutil.disk_usage(d).free
        if d == "/dev/shm":  # (memory: leave 48 GB for everything else)
            try:
                avail = next(int(ln.split()[1]) * 1024 for ln in open("/proc/meminfo") if ln.startswith("MemAvailable"))
                f = min(f, max(avail - (48 << 30), 0))
            except (OSError, StopIteration, ValueError):
                pass
        return f
    for d in cands:
        if free(d) >= 1.5 * bytes_needed:
            return d
    return max(cands, key=free)


def genome_bam_scale(directory, records_per_s: float | None = None, write_budget_s: float = 120.0, depth: float = 60.0):
    """The largest ``scale`` (<= 1) of the whole-genome BAM that ``directory`` holds with room to spare and that the
    writer finishes within ``write_budget_s`` at ``records_per_s`` (measured by the caller on a small file).
    ``FTK_WG_BAM_SCALE`` overrides."""
    import os
    import shutil
    env = os.environ.get("FTK_WG_BAM_SCALE")
    if env:
        return float(env)
    free = shutil.disk_usage(directory).free
    scale = min(1.0, 0.6 * free / genome_bam_bytes(1.0, depth))
    if records_per_s:
        scale = min(scale, write_budget_s * records_per_s / (2 * sum(n_fragments(n, depth) for n in B37_SIZES.values())))
    return max(0.02, round(scale, 3))


def write_random_2bit(path, sizes, seed=SEED_BASE, n_blocks=True):
    """A UCSC ``.2bit`` reference of random bases for ``sizes = {contig: length}`` (little endian, version 0): the
    packed DNA is written as random bytes (T=0 C=1 A=2 G=3, four bases per byte, first base in the high bits), each
    contig gets N blocks over its synthetic telomeres (``synth_gaps``) when ``n_blocks``, no mask blocks.  Written in
    pieces, so a whole genome (775 MB) needs no more than 64 MB of memory.  Returns ``{contig: (N starts, N sizes)}``."""
    import struct
    names = list(sizes)
    head = struct.pack("<IIII", 0x1A412743, 0, len(names), 0)
    index_len = sum(1 + len(n.encode()) + 4 for n in names)
    blocks = {}
    for n in names:
        if n_blocks and sizes[n] > 40_000:
            blocks[n] = ([0, sizes[n] - 10_000], [10_000, 10_000])
        else:
            blocks[n] = ([], [])
    off = len(head) + index_len
    offsets = {}
    for n in names:
        offsets[n] = off
        off += 4 + 4 + 8 * len(blocks[n][0]) + 4 + 4 + (sizes[n] + 3) // 4
    with open(path, "wb") as fh:
        fh.write(head)
        for n in names:
            fh.write(bytes([len(n.encode())]) + n.encode() + struct.pack("<I", offsets[n]))
        for k, n in enumerate(names):
            st, sz = blocks[n]
            fh.write(struct.pack("<II", sizes[n], len(st)))
            fh.write(np.asarray(st, "<u4").tobytes() + np.asarray(sz, "<u4").tobytes())
            fh.write(struct.pack("<II", 0, 0))
            rng = np.random.default_rng(seed + 7919 * k)
            left = (sizes[n] + 3) // 4
            while left > 0:
                m = min(left, 64 << 20)
                fh.write(rng.integers(0, 256, m, dtype=np.uint8).tobytes())
                left -= m
    return blocks


def write_genome_delfi_inputs(directory, sizes, window=_WINDOW, n_blacklist=2000, seed=77):
    """The side files of a whole-genome DELFI run over ``sizes`` (BASELINE config 4): chrom.sizes, the ``window`` bp
    bins file, a blacklist BED (``n_blacklist`` regions spread by contig length, ``synth_blacklist``) and a BED4 gap
    file with every contig's synthetic centromere / telomeres (``synth_gaps``).  Returns the four paths."""
    import os
    total = float(sum(sizes.values()))
    cs, bins, bl, gaps = (os.path.join(directory, f) for f in ("genome.chrom.sizes", "bins.bed", "blacklist.bed", "gaps.bed"))
    with open(cs, "w") as fh:
        fh.write("".join(f"{c}\t{n}\n" for c, n in sizes.items()))
    with open(bins, "w") as fh:
        for c, n in sizes.items():
            ws, we = tiling_windows(n, window)
            fh.write("".join(f"{c}\t{a}\t{b}\n" for a, b in zip(ws.tolist(), we.tolist())))
    with open(bl, "w") as fh:
        for i, (c, n) in enumerate(sizes.items()):
            s, e = synth_blacklist(n, seed + i, max(8, int(n_blacklist * n / total)))
            fh.write("".join(f"{c}\t{a}\t{b}\n" for a, b in zip(s.tolist(), e.tolist())))
    with open(gaps, "w") as fh:
        for c, n in sizes.items():
            c0, c1, telo = synth_gaps(n)
            fh.write(f"{c}\t{c0}\t{c1}\tcentromere\n" + "".join(f"{c}\t{a}\t{b}\ttelomere\n" for a, b in telo))
    return cs, bins, bl, gaps
