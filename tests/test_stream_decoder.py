"""CPU: the streaming decoder (ftk_fragstream_*) hands out, contig by contig, exactly what the
whole-file decoders return -- also when BGZF blocks, text lines and BAM records straddle the
boundaries of the pieces it reads (FTK_STREAM_PIECE makes the pieces small)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from finaletoolkit_amd import _lib as L
from finaletoolkit_amd import bgzf, synth
from tests.helpers import DATA, GOLDEN, ROOT, write_synthetic_bam
from tests.test_abi import _decode


def _stream(path, bam=False, contig=None, threads=3, queued=2):
    lib = L.load()
    s = C.c_void_p()
    rc = lib.ftk_fragstream_open(path.encode(), None if contig is None else contig.encode(), int(bam), threads, queued,
                                 C.byref(s))
    if rc != 0:
        raise RuntimeError((rc, lib.ftk_fragtable_error().decode()))
    out, order = {}, []
    try:
        refs = [(lib.ftk_fragstream_ref_name(s, i).decode(), lib.ftk_fragstream_ref_length(s, i))
                for i in range(lib.ftk_fragstream_n_refs(s))] if bam else []
        while True:
            t = C.c_void_p()
            rc = lib.ftk_fragstream_next(s, C.byref(t))
            if rc != 0:
                raise RuntimeError((rc, lib.ftk_fragtable_error().decode()))
            if not t.value:
                break
            try:
                assert lib.ftk_fragtable_n_contigs(t) == 1
                rows = lib.ftk_fragtable_contig_rows(t, 0)
                name = lib.ftk_fragtable_contig_name(t, 0).decode()
                ps = [C.c_void_p() for _ in range(6)]
                assert lib.ftk_fragtable_columns(t, 0, *[C.byref(p) for p in ps]) == 0
                cols = []
                for p, ct in zip(ps, (C.c_int32, C.c_int32, C.c_uint8, C.c_uint8, C.c_int32, C.c_int32)):
                    cols.append(None if not p.value or rows == 0
                                else np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), (rows,)).copy())
                assert name not in out
                out[name] = (rows, cols, lib.ftk_fragtable_contig_length(t, 0))
                order.append(name)
                out["__bed6__"] = lib.ftk_fragtable_is_bed6(t)
            finally:
                lib.ftk_fragtable_free(t)
    finally:
        lib.ftk_fragstream_close(s)
    return out, order, refs


def _same(got, want):
    names = [k for k in want if not k.startswith("__") and want[k][0] > 0]
    assert sorted(k for k in got if not k.startswith("__")) == sorted(names)
    for k in names:
        assert got[k][0] == want[k][0] and got[k][2] == want[k][2], k
        for a, b in zip(got[k][1], want[k][1]):
            assert (a is None and b is None) or np.array_equal(a, b), k


@pytest.mark.parametrize("name,bam", [("12.3444.b37.frag.gz", False), ("12.3444.b37.frag.bed.gz", False),
                                      ("12.3444.b37.bam", True)])
def test_stream_equals_whole_file_on_fixtures(name, bam):
    path = os.path.join(DATA, name)
    got, order, refs = _stream(path, bam=bam)
    want = _decode(path, bam=bam)
    _same(got, want)
    assert got["__bed6__"] == want["__bed6__"]
    if bam:
        assert len(refs) == 84 and ("12", 133851895) in refs


_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from tests.test_stream_decoder import _stream, _same
from tests.test_abi import _decode
path, bam = sys.argv[1], sys.argv[2] == "1"
for threads in (1, 4):
    got, order, refs = _stream(path, bam=bam, threads=threads, queued=1)
    _same(got, _decode(path, bam=bam))
only, _, _ = _stream(path, bam=bam, contig=sys.argv[3])
assert [k for k in only if not k.startswith("__")] == [sys.argv[3]]
print("ok", order)
"""


def _run_child(path, bam, contig, **extra):
    env = dict(os.environ, FTK_STREAM_PIECE="65536")
    env.update(extra)
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=ROOT), path, "1" if bam else "0", contig], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr
    return r.stdout


def test_stream_small_pieces_text(tmp_path):
    """Four contigs, ~6 MB compressed, read in 64 KB pieces: every kind of boundary straddle occurs."""
    rows = []
    for k, size in enumerate((2_000_000, 900_000, 50_000, 1_500_000)):
        s, e, q, st = synth.synth_contig(size, depth=18.0, seed=40 + k)
        rows.append((f"c{k}", s, e, q, st))
    p = str(tmp_path / "multi.frag.gz")
    bgzf.write_frag_gz(p, rows, level=1)
    out = _run_child(p, False, "c2")
    assert "['c0', 'c1', 'c2', 'c3']" in out
    # a stream's first reads may be short and double up to the piece size (FTK_STREAM_RAMP, off by default; here
    # 64 KB -> 1 MB): the end of the file is "fewer bytes than THAT read asked for", whichever size it had
    assert "['c0', 'c1', 'c2', 'c3']" in _run_child(p, False, "c2", FTK_STREAM_PIECE=str(1 << 20), FTK_STREAM_RAMP="65536")
    assert "['c0', 'c1', 'c2', 'c3']" in _run_child(p, False, "c2", FTK_STREAM_PIECE=str(1 << 20), FTK_STREAM_RAMP="0")


def test_stream_small_pieces_bam(tmp_path):
    rng = np.random.default_rng(9)
    contigs = [("chrA", 3_000_000), ("chrEmpty", 1000), ("chrB", 1_000_000), ("chrC", 400_000)]
    frags = {}
    for name, size in contigs:
        if name == "chrEmpty":
            continue
        n = size // 40
        s = np.sort(rng.integers(0, size - 700, n))
        frags[name] = (s, s + rng.integers(210, 600, n), rng.integers(0, 61, n), rng.integers(0, 2, n).astype(bool))
    p = str(tmp_path / "multi.bam")
    write_synthetic_bam(p, contigs, frags)
    out = _run_child(p, True, "chrB")
    assert "['chrA', 'chrB', 'chrC']" in out
    assert "['chrA', 'chrB', 'chrC']" in _run_child(p, True, "chrB", FTK_STREAM_PIECE=str(1 << 19), FTK_STREAM_RAMP="65536")
    # the record chain walked as many speculative stretches per piece (4 KB: a dozen records each; 100 bytes:
    # shorter than a record, most stretches hold no record start)
    for stretch in ("4096", "100"):
        assert "['chrA', 'chrB', 'chrC']" in _run_child(p, True, "chrB", FTK_BAM_STRETCH=stretch)


def test_bam_stretch_guess_survives_decoy_records(tmp_path):
    """Every record carries, in its quality string, three chained byte patterns that look exactly like BAM
    records of a proper read1 (the stretch-start guesser accepts them).  A thread that enters the chain there
    must be caught by the chain check and its stretch redone: the fragments are those of the real records."""
    import struct
    read = 150
    rng = np.random.default_rng(5)
    n = 6000
    size = 2_000_000
    s = np.sort(rng.integers(1000, size - 2000, n))
    length = rng.integers(160, 500, n)
    rec = np.dtype([("block_size", "<i4"), ("ref", "<i4"), ("pos", "<i4"), ("l_name", "u1"), ("mapq", "u1"),
                    ("bin", "<u2"), ("n_cigar", "<u2"), ("flag", "<u2"), ("l_seq", "<i4"), ("next_ref", "<i4"),
                    ("next_pos", "<i4"), ("tlen", "<i4"), ("name", "S8"), ("cigar", "<u4"),
                    ("seq", "u1", (read // 2,)), ("qual", "u1", (read,))])
    a = np.zeros(n, rec)  # forward read1 records only: fragment = [pos, pos + tlen)
    a["block_size"] = rec.itemsize - 4
    a["l_name"], a["n_cigar"], a["l_seq"], a["cigar"] = 8, 1, read, read << 4
    a["pos"], a["next_pos"], a["tlen"], a["flag"], a["mapq"] = s, s + length - read, length, 99, 60
    a["name"] = np.char.zfill(np.arange(n).astype("U7"), 7).astype("S8")
    # 38 bytes, a valid chain link: block_size 34, a one-letter read name, flags of a proper read1
    decoy = struct.pack("<iiiBBHHHiiii", 34, 0, 777, 2, 60, 0, 0, 99, 0, 0, 900, 222) + b"A\0"
    assert len(decoy) == 38
    q = np.full(read, 30, np.uint8)
    q[20:20 + 114] = np.frombuffer(decoy * 3, np.uint8)
    a["qual"] = q
    text = b"@HD\tVN:1.6\tSO:coordinate\n@SQ\tSN:chrD\tLN:%d\n" % size
    head = b"BAM\1" + struct.pack("<i", len(text)) + text + struct.pack("<i", 1)
    head += struct.pack("<i", 5) + b"chrD\0" + struct.pack("<i", size)
    p = str(tmp_path / "decoy.bam")
    bgzf.write_bgzf(p, head + a.tobytes(), level=1)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests.test_stream_decoder import _stream\n"
            "import pickle\n"
            "got, order, _ = _stream(%r, bam=True, threads=16)\n"
            "pickle.dump(got, open(%r, 'wb'))\n") % (ROOT, p, str(tmp_path / "got.pkl"))
    import pickle
    import re
    for stretch in ("1000", "4096", "70000"):
        r = subprocess.run([sys.executable, "-c", code], check=True, capture_output=True, text=True,
                           env=dict(os.environ, FTK_BAM_STRETCH=stretch, FTK_STREAM_PIECE=str(1 << 16),
                                    FTK_DECODE_TIMING="1"))
        m = re.search(r"(\d+) stretches of the record chain, (\d+) redone", r.stderr)
        assert m and int(m.group(1)) >= 16 and int(m.group(2)) >= 1, r.stderr[-500:]  # decoys were entered
        got = pickle.load(open(tmp_path / "got.pkl", "rb"))
        rows, cols, length_ = got["chrD"]
        assert rows == n and length_ == size, (stretch, rows)
        assert np.array_equal(cols[0], s) and np.array_equal(cols[1], s + length)


def test_stream_errors(tmp_path):
    with pytest.raises(RuntimeError):
        _stream(str(tmp_path / "absent.gz"))
    junk = tmp_path / "junk.gz"
    junk.write_bytes(b"this is not gzip")
    with pytest.raises(RuntimeError):
        _stream(str(junk))
    # a contig that comes back after another one: not coordinate-sorted
    s, e, q, st = synth.synth_contig(200_000, depth=5.0, seed=1)
    p = str(tmp_path / "unsorted.frag.gz")
    bgzf.write_frag_gz(p, [("a", s, e, q, st), ("b", s, e, q, st), ("a", s, e, q, st)], level=1)
    with pytest.raises(RuntimeError) as ei:
        _stream(p)
    assert ei.value.args[0][0] == L.FTK_ERR_UNSORTED
    # truncated BGZF file
    data = open(os.path.join(GOLDEN, "synth.frag.gz"), "rb").read()
    cut = tmp_path / "cut.frag.gz"
    cut.write_bytes(data[: len(data) // 2])
    with pytest.raises(RuntimeError):
        _stream(str(cut))
    # closing a stream that was never drained must not hang
    lib = L.load()
    h = C.c_void_p()
    assert lib.ftk_fragstream_open(os.path.join(GOLDEN, "synth.frag.gz").encode(), None, 0, 2, 1, C.byref(h)) == 0
    lib.ftk_fragstream_close(h)


def test_stream_plain_gzip_falls_back(tmp_path):
    import gzip
    p = str(tmp_path / "plain.frag.gz")
    with gzip.open(p, "wt") as fh:
        fh.write("1\t10\t200\t60\t+\n1\t50\t260\t30\t-\n2\t5\t100\t9\t+\n")
    got, order, _ = _stream(p)
    assert order == ["1", "2"] and got["1"][1][0].tolist() == [10, 50] and got["2"][1][2].tolist() == [9]


def test_single_contig_requests_use_the_index(tmp_path):
    """With a tabix / BAI index a single-contig stream reads only that contig's blocks: same rows as a scan
    of the whole file, contigs without rows or unknown to the file give nothing, a stub index falls back."""
    rows = []
    for k, size in enumerate((1_200_000, 300_000, 40_000, 900_000)):
        s, e, q, st = synth.synth_contig(size, depth=15.0, seed=60 + k)
        rows.append((f"c{k}", s, e, q, st))
    pi, ps = str(tmp_path / "idx.frag.gz"), str(tmp_path / "stub.frag.gz")
    bgzf.write_frag_gz(pi, rows, level=1, with_index=True)
    bgzf.write_frag_gz(ps, rows, level=1)
    want = _decode(ps)
    for c in ("c0", "c1", "c2", "c3"):
        for path in (pi, ps):
            got, order, _ = _stream(path, contig=c)
            assert order == [c]
            for a, b in zip(got[c][1][:4], want[c][1][:4]):
                assert np.array_equal(a, b), (path, c)
    assert _stream(pi, contig="c9")[1] == [] and _stream(ps, contig="c9")[1] == []
    # BAM + BAI, incl. a contig of the header without reads and one the header does not know
    rng = np.random.default_rng(3)
    contigs = [("chrA", 800_000), ("chrEmpty", 1000), ("chrB", 500_000), ("chrC", 100_000)]
    frags = {}
    for name, size in contigs:
        if name == "chrEmpty":
            continue
        n = size // 50
        s = np.sort(rng.integers(0, size - 700, n))
        frags[name] = (s, s + rng.integers(210, 600, n), rng.integers(0, 61, n), rng.integers(0, 2, n).astype(bool))
    bi, bs = str(tmp_path / "idx.bam"), str(tmp_path / "stub.bam")
    write_synthetic_bam(bi, contigs, frags, index=True)
    write_synthetic_bam(bs, contigs, frags, index=False)
    wantb = _decode(bs, bam=True)
    for c in ("chrA", "chrB", "chrC"):
        for path in (bi, bs):
            got, order, refs = _stream(path, bam=True, contig=c)
            assert order == [c] and len(refs) == 4
            for a, b in zip(got[c][1], wantb[c][1]):
                assert np.array_equal(a, b), (path, c)
    for path in (bi, bs):
        assert _stream(path, bam=True, contig="chrEmpty")[1] == []
        assert _stream(path, bam=True, contig="chrZ")[1] == []
    # the fixtures' real htslib indexes
    assert _stream(os.path.join(DATA, "12.3444.b37.frag.gz"), contig="12")[0]["12"][0] == 17
    assert _stream(os.path.join(DATA, "12.3444.b37.bam"), bam=True, contig="12")[0]["12"][0] == 17
    assert _stream(os.path.join(DATA, "12.3444.b37.bam"), bam=True, contig="1")[1] == []


def _int_rule(f):
    """The decoder's integer field rule (csrc/ftk_decode.cpp parse_int): blanks around, one sign, digits."""
    f = f.strip(" \r")
    body = f[1:] if f[:1] in "+-" else f
    if not body or not body.isascii() or not body.isdigit():
        raise ValueError(f)
    v = int(f)
    if abs(v) > (1 << 40):
        raise ValueError(f)
    return v


def _rows_by_rule(text, bed6):
    mq_col, st_col = (4, 5) if bed6 else (3, 4)
    out = {}
    for line in text.split("\n"):
        if line.endswith("\r"):
            line = line[:-1]
        if not line or line.startswith("#"):
            continue
        p = line.split("\t")
        if len(p) <= st_col:
            continue
        try:
            s, t, m = _int_rule(p[1]), _int_rule(p[2]), _int_rule(p[mq_col])
        except ValueError:
            continue
        if s < 0 or t < 0 or s > 2**31 - 1 or t > 2**31 - 1 or m < 0:
            continue
        out.setdefault(p[0], []).append((s, t, min(m, 255), 1 if "+" in p[st_col] else 0))
    return out


@pytest.mark.parametrize("bed6", [False, True])
def test_text_rows_plain_fast_path_equals_general_rules(tmp_path, bed6):
    """Mostly plain rows (the one-pass fast path) with every kind of odd row mixed in: the decoded columns
    must be what the field rules give row by row, whole-file and streamed, at several thread counts."""
    rng = np.random.default_rng(77 + bed6)
    odd = [
        lambda c, s, e, q, st: [c, " %d" % s, str(e), str(q), st],
        lambda c, s, e, q, st: [c, str(s), "+%d" % e, str(q), st],
        lambda c, s, e, q, st: [c, "-%d" % s, str(e), str(q), st],
        lambda c, s, e, q, st: [c, str(s), str(e), "6x", st],
        lambda c, s, e, q, st: [c, "00000000%d" % (s % 1000), str(e), str(q), st],     # 11 digits, small value
        lambda c, s, e, q, st: [c, "123456789012345", str(e), str(q), st],
        lambda c, s, e, q, st: [c, "2147483648", str(e), str(q), st],
        lambda c, s, e, q, st: [c, str(s), "2147483647", str(q), st],
        lambda c, s, e, q, st: [c, str(s), str(e), "300", st],
        lambda c, s, e, q, st: [c, str(s), str(e), " %d " % q, st],
        lambda c, s, e, q, st: [c, str(s), str(e), str(q), "+-"],
        lambda c, s, e, q, st: [c, str(s), str(e), str(q), "."],
        lambda c, s, e, q, st: [c, str(s), str(e), str(q), st, "extra", "more", "cols", "here"],
        lambda c, s, e, q, st: [c, str(s), str(e), str(q)],
        lambda c, s, e, q, st: [c, "", str(e), str(q), st],
        lambda c, s, e, q, st: [c, str(s), str(e), str(q), ""],
        lambda c, s, e, q, st: ["#" + c, str(s), str(e), str(q), st],
        lambda c, s, e, q, st: [""],
        lambda c, s, e, q, st: [c + "x", str(s), str(e), str(q), st],                 # a one-row run of another name
    ]
    lines = []
    for c in ("chr1", "chr10", "2"):
        n = 40_000
        s = np.sort(rng.integers(0, 50_000_000, n))
        e = s + rng.integers(1, 600, n)
        q = rng.integers(0, 61, n)
        st = rng.integers(0, 2, n)
        pick = rng.random(n)
        for i in range(n):
            f = [c, str(s[i]), str(e[i]), str(q[i]), "+" if st[i] else "-"]
            if pick[i] < 0.05:
                f = odd[int(rng.integers(len(odd)))](c, int(s[i]), int(e[i]), int(q[i]), f[4])
            if bed6 and len(f) >= 4:
                f = f[:3] + ["frag %d" % i] + f[3:]
            lines.append("\t".join(f) + ("\r" if pick[i] > 0.97 else ""))
    # the first data row fixes the layout (io/alignment.py:143-156): make it a plain one
    first = ["chr1", "5", "170", "name", "60", "+"] if bed6 else ["chr1", "5", "170", "60", "+"]
    text = "\t".join(first) + "\n" + "\n".join(lines)      # no newline after the last row
    want = _rows_by_rule(text, bed6)
    # names such as "chr1x" come back as separate one-row runs between two runs of "chr1": not a sorted file for the
    # streaming decoder, so compare run-insensitive through the whole-file decoder and stream a file without them
    p = str(tmp_path / "rows.frag.gz")
    bgzf.write_bgzf(p, text.encode(), level=1)
    for threads in (1, 4):
        got = _decode(p, threads=threads)
        assert got["__bed6__"] == int(bed6)
        assert sorted(k for k in got if not k.startswith("__")) == sorted(want)
        for c, rows in want.items():
            a = np.array(rows, dtype=np.int64)
            assert got[c][0] == len(rows), c
            for k in range(4):
                assert np.array_equal(got[c][1][k].astype(np.int64), a[:, k]), (c, k)
    sorted_text = "\n".join(ln for ln in text.split("\n") if not ln.split("\t")[0].endswith("x")) + "\n"
    want = _rows_by_rule(sorted_text, bed6)
    p2 = str(tmp_path / "rows_sorted.frag.gz")
    bgzf.write_bgzf(p2, sorted_text.encode(), level=1)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests.test_stream_decoder import _stream\n"
            "import numpy as np, pickle\n"
            "got, order, _ = _stream(%r, threads=3)\n"
            "pickle.dump({k: v for k, v in got.items()}, open(%r, 'wb'))\n") % (ROOT, p2, str(tmp_path / "got.pkl"))
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, FTK_STREAM_PIECE=str(1 << 16)))
    import pickle
    got = pickle.load(open(tmp_path / "got.pkl", "rb"))
    assert sorted(k for k in got if not k.startswith("__")) == sorted(want)
    for c, rows in want.items():
        a = np.array(rows, dtype=np.int64)
        assert got[c][0] == len(rows), c
        for k in range(4):
            assert np.array_equal(got[c][1][k].astype(np.int64), a[:, k]), (c, k)
