"""GPU: the streaming text decoder with the row parser on the device (ftk_fragstream_open_device,
csrc/ftk_textparse.hip) hands out exactly the rows of the host decoders - on the fixtures, on multi-contig
files read in 64 KB pieces (contig runs inside a piece, lines straddling pieces), on files full of odd rows
(those pieces go through the host's field-rule parser), with CRLF line ends, without a final newline, and for
single-contig requests through the index."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from finaletoolkit_amd import _lib as L
from finaletoolkit_amd import bgzf, synth
from tests.helpers import DATA, GOLDEN, ROOT

pytestmark = pytest.mark.gpu


def _stream_device(path, contig=None, threads=3, queued=2):
    lib = L.load()
    s = C.c_void_p()
    rc = lib.ftk_fragstream_open_device(0, path.encode(), None if contig is None else contig.encode(), 0, threads, queued,
                                        C.byref(s))
    if rc != 0:
        raise RuntimeError((rc, lib.ftk_fragtable_error().decode()))
    out, order, n_dev = {}, [], 0
    try:
        while True:
            t = C.c_void_p()
            rc = lib.ftk_fragstream_next(s, C.byref(t))
            if rc != 0:
                raise RuntimeError((rc, lib.ftk_fragtable_error().decode()))
            if not t.value:
                break
            try:
                rows = lib.ftk_fragtable_contig_rows(t, 0)
                name = lib.ftk_fragtable_contig_name(t, 0).decode()
                n_dev += lib.ftk_fragtable_is_device(t, 0)
                cols = [np.empty(rows, np.int32), np.empty(rows, np.int32), np.empty(rows, np.uint8), np.empty(rows, np.uint8)]
                assert lib.ftk_fragtable_columns_to_host(t, 0, *[c.ctypes.data_as(C.c_void_p) for c in cols]) == 0
                assert name not in out
                out[name] = (rows, cols)
                order.append(name)
                out["__bed6__"] = lib.ftk_fragtable_is_bed6(t)
            finally:
                lib.ftk_fragtable_free(t)
    finally:
        lib.ftk_fragstream_close(s)
    return out, order, n_dev


def _whole(path, contig=None):
    from tests.test_abi import _decode
    return _decode(path, contig=contig)


def _same(got, want):
    names = [k for k in want if not k.startswith("__") and want[k][0] > 0]
    assert sorted(k for k in got if not k.startswith("__")) == sorted(names)
    assert got.get("__bed6__", want["__bed6__"]) == want["__bed6__"]
    for k in names:
        assert got[k][0] == want[k][0], k
        for a, b in zip(got[k][1], want[k][1][:4]):
            assert np.array_equal(a, b), k


_CHILD = """
import sys
sys.path.insert(0, {root!r})
from tests.test_gpu_device_parse import _stream_device, _whole, _same
path = sys.argv[1]
got, order, n_dev = _stream_device(path, threads=int(sys.argv[2]))
_same(got, _whole(path))
if len(sys.argv) > 3:
    only, _, _ = _stream_device(path, contig=sys.argv[3])
    assert [k for k in only if not k.startswith("__")] == [sys.argv[3]]
    _same(only, _whole(path, contig=sys.argv[3]))
print("ok", order, n_dev)
"""


def _child(path, threads=3, contig=None, **env):
    args = [sys.executable, "-c", _CHILD.format(root=ROOT), path, str(threads)] + ([contig] if contig else [])
    r = subprocess.run(args, env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout + r.stderr[-3000:]
    return r.stdout + r.stderr


@pytest.mark.parametrize("name", ["12.3444.b37.frag.gz", "12.3444.b37.frag.bed.gz"])
def test_device_rows_equal_host_rows_on_fixtures(name):
    path = os.path.join(DATA, name)
    got, order, n_dev = _stream_device(path)
    _same(got, _whole(path))
    assert n_dev == len(order) == 1
    got, order, _ = _stream_device(os.path.join(GOLDEN, "synth.frag.gz"), threads=4)
    _same(got, _whole(os.path.join(GOLDEN, "synth.frag.gz")))


def test_device_rows_multi_contig_small_pieces(tmp_path):
    rows = []
    for k, size in enumerate((2_000_000, 900_000, 50_000, 1_500_000)):
        s, e, q, st = synth.synth_contig(size, depth=18.0, seed=40 + k)
        rows.append((f"c{k}", s, e, q, st))
    p = str(tmp_path / "multi.frag.gz")
    bgzf.write_frag_gz(p, rows, level=1)
    out = _child(p, contig="c2", FTK_STREAM_PIECE="65536", FTK_DECODE_TIMING="1")
    assert "['c0', 'c1', 'c2', 'c3']" in out
    import re
    m = re.search(r"(\d+) pieces parsed on the device, (\d+) by the host", out)
    assert m and int(m.group(1)) > 20 and int(m.group(2)) == 0, out[-600:]
    # whole pieces (one 48 MB piece holds the file) and the switch that keeps the rows on the host
    assert "['c0', 'c1', 'c2', 'c3']" in _child(p, threads=8)
    assert "] 0" in _child(p, FTK_DEVICE_PARSE="0").splitlines()[0]  # no device table handed out


def test_device_contigs_growing_over_many_pieces(tmp_path):
    """Contigs of 2.5 M rows arriving in ~150 k-row pieces: the device block of a contig is regrown (1 M -> 2 M ->
    4 M rows) while it fills, and blocks are recycled from contig to contig."""
    rows = []
    for k in range(3):
        s, e, q, st = synth.synth_contig(60_000_000, seed=70 + k, n=2_500_037 - 400_011 * k)
        rows.append((f"g{k}", s, e, q, st))
    p = str(tmp_path / "grow.frag.gz")
    bgzf.write_frag_gz(p, rows, level=1)
    out = _child(p, threads=8, FTK_STREAM_PIECE=str(1 << 20))
    assert "['g0', 'g1', 'g2']" in out
    # the first reads short, doubling up to the piece size (FTK_STREAM_RAMP, off by default): 64 KB -> 1 MB
    assert "['g0', 'g1', 'g2']" in _child(p, threads=8, FTK_STREAM_PIECE=str(1 << 20), FTK_STREAM_RAMP="65536")
    # one piece per file: a contig's first block is sized by an arbitrary row count (not a multiple of 64)
    assert "['g0', 'g1', 'g2']" in _child(p, threads=8)
    # page-locked blocks the old way (hipHostMalloc instead of a registered mapping: the fall-back of pinned_map)
    assert "['g0', 'g1', 'g2']" in _child(p, threads=8, FTK_PINNED_VIA="malloc")


@pytest.mark.parametrize("bed6", [False, True])
def test_device_rows_with_odd_rows_crlf_and_no_final_newline(tmp_path, bed6):
    """Plain pieces stay on the device, pieces with odd rows go through the host's field rules; the rows are
    the rules' rows either way (the same generator as tests/test_stream_decoder.py)."""
    from tests.test_stream_decoder import _rows_by_rule
    rng = np.random.default_rng(5 + bed6)
    lines = []
    for c in ("chr1", "chr10", "2"):
        n = 60_000
        s = np.sort(rng.integers(0, 50_000_000, n))
        e = s + rng.integers(1, 600, n)
        q = rng.integers(0, 300, n)
        st = rng.integers(0, 2, n)
        pick = rng.random(n)
        # odd rows only in the middle third of every contig: the pieces before and after are plain
        for i in range(n):
            f = [c, str(s[i]), str(e[i]), str(q[i]), "+" if st[i] else "-"]
            if n // 3 < i < 2 * n // 3 and pick[i] < 0.01:
                f = [[c, " %d" % s[i], str(e[i]), str(q[i]), f[4]], [c, str(s[i]), "+%d" % e[i], str(q[i]), f[4]],
                     [c, str(s[i]), str(e[i]), "6x", f[4]], [c, "2147483648", str(e[i]), str(q[i]), f[4]],
                     [c, str(s[i]), str(e[i]), str(q[i]), "+-"], [c, str(s[i]), str(e[i]), str(q[i])],
                     ["#" + c, str(s[i]), str(e[i]), str(q[i]), f[4]], [""],
                     [c, str(s[i]), str(e[i]), str(q[i]), f[4], "extra"]][int(rng.integers(9))]
            if bed6 and len(f) >= 4:
                f = f[:3] + ["frag %d" % i] + f[3:]
            lines.append("\t".join(f) + ("\r" if pick[i] > 0.9 else ""))
    first = ["chr1", "5", "170", "name", "60", "+"] if bed6 else ["chr1", "5", "170", "60", "+"]
    text = "\t".join(first) + "\n" + "\n".join(lines)  # no newline after the last row
    want = _rows_by_rule(text, bed6)
    p = str(tmp_path / "rows.frag.gz")
    bgzf.write_bgzf(p, text.encode(), level=1)
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from tests.test_gpu_device_parse import _stream_device\n"
            "import pickle\n"
            "got, order, n_dev = _stream_device(%r, threads=3)\n"
            "pickle.dump(got, open(%r, 'wb'))\n") % (ROOT, p, str(tmp_path / "got.pkl"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True,
                       env=dict(os.environ, FTK_STREAM_PIECE=str(1 << 16), FTK_DECODE_TIMING="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    import pickle
    import re
    got = pickle.load(open(tmp_path / "got.pkl", "rb"))
    assert got["__bed6__"] == int(bed6)
    assert sorted(k for k in got if not k.startswith("__")) == sorted(want)
    for c, rows in want.items():
        a = np.array(rows, dtype=np.int64)
        assert got[c][0] == len(rows), c
        for k in range(4):
            assert np.array_equal(got[c][1][k].astype(np.int64), a[:, k]), (c, k)
    m = re.search(r"(\d+) pieces parsed on the device, (\d+) by the host", r.stderr)
    assert m and int(m.group(1)) >= 3 and int(m.group(2)) >= 3, r.stderr[-600:]  # both paths were taken


def test_sources_opened_from_text_files_use_the_device_rows(tmp_path):
    """open_source / stream_source (the reference-shaped API's way in) on a text file: fragments identical to
    an upload of the host-decoded columns."""
    from finaletoolkit_amd import source
    s, e, q, st = synth.synth_contig(3_000_000, depth=12.0, seed=9)
    p = str(tmp_path / "one.frag.gz")
    bgzf.write_frag_gz(p, [("chrT", s, e, q, st)], level=1)
    eng = source.get_engine()
    src = source.open_source(p)
    ws, we = synth.tiling_windows(3_000_000, 100_000)
    got = eng.window_features(src.require("chrT"), ws, we, 30, hist=(0, 600))
    eng.load_contig("ref:chrT", s, e, q, st)
    want = eng.window_features("ref:chrT", ws, we, 30, hist=(0, 600))
    assert np.array_equal(got["coverage"], want["coverage"]) and np.array_equal(got["hist"], want["hist"])
    gs, ge, gq, gst = eng.frag_select(src.require("chrT"), None, None, 0, None, None, "any")
    assert np.array_equal(gs, s) and np.array_equal(ge, e) and np.array_equal(gq, q) and np.array_equal(gst, st)
    eng.release("ref:chrT")
    source.close_all()


def test_device_stream_errors_and_early_close(tmp_path):
    """Closing a device-mode stream before it is drained, a truncated file and an unsorted file: clean errors,
    no hang, and the next stream works (the buffer sets and device blocks go back to their pools)."""
    lib = L.load()
    rows = []
    for k in range(4):
        s, e, q, st = synth.synth_contig(2_000_000, depth=10.0, seed=20 + k)
        rows.append((f"e{k}", s, e, q, st))
    good = str(tmp_path / "good.frag.gz")
    bgzf.write_frag_gz(good, rows, level=1)
    # (a) close after the first contig / without reading anything (the producer is waiting for queue space)
    for take in (0, 1):
        s = C.c_void_p()
        assert lib.ftk_fragstream_open_device(0, good.encode(), None, 0, 4, 1, C.byref(s)) == 0
        for _ in range(take):
            t = C.c_void_p()
            assert lib.ftk_fragstream_next(s, C.byref(t)) == 0 and t.value
            lib.ftk_fragtable_free(t)
        lib.ftk_fragstream_close(s)
    # (b) truncated in the middle of a BGZF block
    raw = open(good, "rb").read()
    cut = str(tmp_path / "cut.frag.gz")
    open(cut, "wb").write(raw[:len(raw) // 2])
    with pytest.raises(RuntimeError):
        _stream_device(cut)
    # (c) a contig that comes back after another one
    unsorted = str(tmp_path / "unsorted.frag.gz")
    bgzf.write_frag_gz(unsorted, [rows[0], rows[1], (rows[0][0],) + rows[2][1:]], level=1)
    with pytest.raises(RuntimeError) as ei:
        _stream_device(unsorted)
    assert "two separate runs" in str(ei.value)
    got, order, n_dev = _stream_device(good)
    assert order == ["e0", "e1", "e2", "e3"] and n_dev == 4
    _same(got, _whole(good))


def test_one_row_per_bgzf_block(tmp_path):
    """A valid BGZF file of many tiny blocks: the compressed piece is LARGER than its text (26 bytes of container
    per 20-byte row), and the compressed bytes are staged in the text buffer on their way up (round-2 advisor
    finding: that buffer was sized from the text alone).  Also blocks with nothing in them between the rows."""
    from finaletoolkit_amd.bgzf import _EOF, _block
    rng = np.random.default_rng(77)
    n = 16_000
    s = np.sort(rng.integers(0, 40_000_000, n))
    e = s + rng.integers(30, 600, n)
    q = rng.integers(0, 61, n)
    rows = ["tiny\t%d\t%d\t%d\t%s\n" % (s[i], e[i], q[i], "+-"[i & 1]) for i in range(n)]
    p = str(tmp_path / "tiny_blocks.frag.gz")
    with open(p, "wb") as fh:
        for i, r in enumerate(rows):
            fh.write(_block(r.encode(), 1))
            if i % 97 == 0:
                fh.write(_EOF)  # an empty block in the middle of the file is legal
        fh.write(_EOF)
    assert os.path.getsize(p) > sum(map(len, rows))
    got, order, n_dev = _stream_device(p, threads=4)
    assert order == ["tiny"] and n_dev == 1 and got["tiny"][0] == n
    assert np.array_equal(got["tiny"][1][0], s) and np.array_equal(got["tiny"][1][1], e)
    assert np.array_equal(got["tiny"][1][2], q) and np.array_equal(got["tiny"][1][3], (np.arange(n) & 1) == 0)
    out = _child(p, threads=4, FTK_STREAM_PIECE=str(1 << 17))  # the same over several pieces
    assert "['tiny']" in out


def test_a_block_claiming_more_than_64k_is_refused(tmp_path):
    import struct
    from finaletoolkit_amd.bgzf import _EOF, _block
    blk = bytearray(_block(b"c\t1\t200\t60\t+\n" * 100, 6))
    blk[-4:] = struct.pack("<I", 70_000)  # ISIZE beyond what a BGZF block may hold
    p = str(tmp_path / "isize.frag.gz")
    open(p, "wb").write(bytes(blk) + _EOF)
    with pytest.raises(RuntimeError):
        _stream_device(p)


@pytest.mark.parametrize("where", ["header", "middle"])
def test_lines_longer_than_the_device_carry(tmp_path, where):
    """A comment line of 100-300 KB (longer than the 64 KB an unfinished line may be carried on the device): the
    stream falls back to the host-inflate pass for the file instead of failing (round-2 advisor finding), hands
    out every contig exactly once and the rows are those of the whole-file decoder."""
    rows = []
    for k in range(3):
        s, e, q, st = synth.synth_contig(900_000, depth=15.0, seed=300 + k)
        rows.append((f"L{k}", s, e, q, st))
    text = []
    for name, s, e, q, st in rows:
        lines = ["%s\t%d\t%d\t%d\t%s\n" % (name, s[i], e[i], q[i], "+" if st[i] else "-") for i in range(len(s))]
        if where == "middle" and name == "L1":
            lines.insert(len(lines) // 2, "#" + "x" * 300_000 + "\n")
        text.append("".join(lines))
    head = "#" + "h" * 100_000 + "\n" if where == "header" else ""
    p = str(tmp_path / f"long_{where}.frag.gz")
    bgzf.write_bgzf(p, (head + "".join(text)).encode(), level=1)
    open(p + ".tbi", "ab").close()
    for piece in (1 << 16, 1 << 20, 48 << 20):
        out = _child(p, threads=4, FTK_STREAM_PIECE=str(piece))
        assert "['L0', 'L1', 'L2']" in out, out


def _many_contig_file(tmp_path, name="hs.frag.gz", rows_per_contig=60_000, n_contigs=5, seed=9):
    rng = np.random.default_rng(seed)
    lines = []
    for c in range(n_contigs):
        s = np.sort(rng.integers(0, 40_000_000, rows_per_contig))
        e = s + rng.integers(30, 600, rows_per_contig)
        q = rng.integers(0, 61, rows_per_contig)
        t = rng.integers(0, 2, rows_per_contig)
        lines += [f"c{c}\t{a}\t{b}\t{m}\t{'+' if k else '-'}\n" for a, b, m, k in zip(s.tolist(), e.tolist(), q.tolist(), t.tolist())]
    p = str(tmp_path / name)
    bgzf.write_bgzf(p, "".join(lines).encode())
    return p


@pytest.mark.parametrize("mode", [dict(FTK_TEXT_HOST_SHARE="2"), dict(FTK_TEXT_HOST_SHARE="1"), dict(FTK_TEXT_HOST_SHARE="0"),
                                  dict(FTK_TEXT_HOST_SHARE="3", FTK_TEXT_DIRECT_UP="0"), dict(FTK_TEXT_DIRECT_UP="0")])
def test_text_pieces_shared_with_the_host_threads(tmp_path, mode):
    """A text stream of ~40 pieces of 64 KB: every second / third piece (or whichever the idle host threads take)
    inflated by the host threads beside the GPU with the backs of the pieces behind it deferred, compressed bytes sent
    up from resting read buffers or staged by a copy - the rows are those of the whole-file decoder in every mode."""
    p = _many_contig_file(tmp_path)
    out = _child(p, threads=8, contig="c3", FTK_STREAM_PIECE="65536", FTK_DECODE_TIMING="1", **mode)
    assert "['c0', 'c1', 'c2', 'c3', 'c4']" in out
    share = mode.get("FTK_TEXT_HOST_SHARE", "1")
    took = [int(line.split(";")[1].split()[0]) for line in out.splitlines() if "inflated by the host threads" in line]
    assert took, out[-2000:]
    if os.environ.get("FTK_DEVICE_INFLATE") == "0":
        assert took[0] == 0, took  # (the suite's inner run with the inflate on the host threads: nothing to share)
    elif share in ("2", "3"):
        assert took[0] >= 10, took  # (the whole-file pass: every 2nd / 3rd of ~40 pieces)
    if share == "0":
        assert took[0] == 0, took


@pytest.mark.parametrize("share", ["0", "2"])
def test_text_stream_with_a_damaged_block_is_an_error(tmp_path, share):
    """One flipped payload byte in the middle of a fragment file: whichever side inflates that piece - the GPU (CRC
    kernel) or the host threads (libdeflate + CRC) - the stream ends with a format error."""
    good = _many_contig_file(tmp_path, "good.frag.gz", rows_per_contig=20_000, n_contigs=3)
    image = bytearray(open(good, "rb").read())
    off, blocks = 0, []
    while off < len(image):
        bs = int.from_bytes(image[off + 16:off + 18], "little") + 1
        blocks.append((off, bs))
        off += bs
    env = dict(FTK_STREAM_PIECE="65536", FTK_TEXT_HOST_SHARE=share)
    _child(good, threads=8, **env)
    for which in (len(blocks) // 2, len(blocks) // 2 + 1, len(blocks) // 2 + 2, len(blocks) // 2 + 3):  # (pieces of either side)
        img = bytearray(image)
        o, bs = blocks[which]
        img[o + 18 + (bs - 26) // 2] ^= 0x5A
        bad = str(tmp_path / f"bad{which}.frag.gz")
        open(bad, "wb").write(bytes(img))
        code = ("import sys\nsys.path.insert(0, %r)\nfrom tests.test_gpu_device_parse import _stream_device\n"
                "try:\n    _stream_device(sys.argv[1], threads=8)\n    print('decoded')\n"
                "except RuntimeError as e:\n    print('error', e)\n" % ROOT)
        r = subprocess.run([sys.executable, "-c", code, bad], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.startswith("error"), (which, r.stdout, r.stderr[-1500:])
        assert str(L.FTK_ERR_FORMAT) in r.stdout, r.stdout


def _stream_region(path, contig, start, stop, threads=4):
    lib = L.load()
    s = C.c_void_p()
    rc = lib.ftk_fragstream_open_region(0, path.encode(), contig.encode(), start, stop, 0, threads, 1, C.byref(s))
    if rc != 0:
        raise RuntimeError((rc, lib.ftk_fragtable_error().decode()))
    tables = []
    try:
        while True:
            t = C.c_void_p()
            rc = lib.ftk_fragstream_next(s, C.byref(t))
            if rc != 0:
                raise RuntimeError((rc, lib.ftk_fragtable_error().decode()))
            if not t.value:
                break
            try:
                rows = lib.ftk_fragtable_contig_rows(t, 0)
                name = lib.ftk_fragtable_contig_name(t, 0).decode()
                cols = [np.empty(rows, np.int32), np.empty(rows, np.int32), np.empty(rows, np.uint8), np.empty(rows, np.uint8)]
                assert lib.ftk_fragtable_columns_to_host(t, 0, *[c.ctypes.data_as(C.c_void_p) for c in cols]) == 0
                tables.append((name, cols))
            finally:
                lib.ftk_fragtable_free(t)
    finally:
        lib.ftk_fragstream_close(s)
    return tables


_REGION_CHILD = """
import sys
import numpy as np
sys.path.insert(0, {root!r})
from tests.test_gpu_device_parse import _stream_region, _whole
path = sys.argv[1]
whole = _whole(path)
report = []
for spec in sys.argv[2:]:
    contig, a, b = spec.split(":")
    a, b = int(a), int(b)
    got = _stream_region(path, contig, a, b)
    n_all, (S, E, Q, T) = whole[contig][0], whole[contig][1][:4]
    need = np.nonzero((S < b) & (E > a))[0]
    if not got:
        assert len(need) == 0, (spec, len(need))
        report.append((spec, 0, n_all))
        continue
    assert len(got) == 1 and got[0][0] == contig, (spec, [g[0] for g in got])
    s, e, q, t = got[0][1]
    # a contiguous slice of the contig's rows ...
    i0 = int(np.searchsorted(S, s[0], side="left"))
    while i0 < n_all and not (S[i0] == s[0] and E[i0] == e[0] and Q[i0] == q[0] and T[i0] == t[0]):
        i0 += 1
    assert i0 + len(s) <= n_all and np.array_equal(S[i0:i0 + len(s)], s) and np.array_equal(E[i0:i0 + len(s)], e) and \\
        np.array_equal(Q[i0:i0 + len(s)], q) and np.array_equal(T[i0:i0 + len(s)], t), spec
    # ... that holds every row overlapping the region
    if len(need):
        assert i0 <= need[0] and need[-1] < i0 + len(s), (spec, i0, len(s), int(need[0]), int(need[-1]))
    report.append((spec, len(s), n_all))
print("ok", report)
"""


def test_region_streams_hold_every_overlapping_row(tmp_path):
    """ftk_fragstream_open_region on a three-contig file with a full tabix index (16 kb linear index): regions at the
    contig's ends, in its middle, across a stretch without rows, one base wide, beyond the last row, and - the case the
    linear index cannot promise - rows up to 60 kb long, which hide where the rows behind the region begin (the
    stream reads on until a parsed row starts behind the region).  Each table is a contiguous slice of the contig's
    rows with every overlapping row in it, and a small region reads a small part of the contig."""
    rng = np.random.default_rng(21)
    rows = []
    for name, n, size, long_rows in (("c1", 400_000, 60_000_000, False), ("c2", 300_000, 40_000_000, True), ("c3", 50_000, 5_000_000, False)):
        s = np.sort(rng.integers(0, size, n))
        if name == "c1":
            s = s[(s < 20_000_000) | (s > 23_000_000)]  # a stretch without rows
        e = s + rng.integers(30, 600, len(s))
        if long_rows:
            pick = rng.random(len(s)) < 0.002
            e[pick] = s[pick] + rng.integers(20_000, 60_000, int(pick.sum()))
        rows.append((name, s, e, rng.integers(0, 61, len(s)), rng.integers(0, 2, len(s))))
    # c4: two rows 5 Mb long - the linear index's windows behind a region inside them point at THEM, far in front of the
    # rows that start just before the region's end: the hint falls short and the stream has to read on
    s = np.sort(np.concatenate([rng.integers(0, 20_000_000, 300_000), [2_000_000, 9_000_000]]))
    e = s + rng.integers(30, 600, len(s))
    e[np.searchsorted(s, 2_000_000)] = 7_000_000
    e[np.searchsorted(s, 9_000_000)] = 14_000_000
    rows.append(("c4", s, e, rng.integers(0, 61, len(s)), rng.integers(0, 2, len(s))))
    p = str(tmp_path / "reg.frag.gz")
    bgzf.write_frag_gz(p, rows, with_index=True)
    specs = ["c4:6000000:6100000", "c4:13000000:13500000", "c4:1990000:2010000", "c1:0:100000", "c1:30000000:31000000", "c1:19990000:23010000", "c1:21000000:22000000", "c1:59000000:70000000",
             "c1:12345678:12345679", "c2:0:1", "c2:10000000:10100000", "c2:25000000:25016384", "c2:39000000:40000000",
             "c2:5000000:30000000", "c3:0:5000000", "c3:4999000:5000000", "c1:70000000:80000000"]
    for env in (dict(), dict(FTK_STREAM_PIECE=str(1 << 16))):
        r = subprocess.run([sys.executable, "-c", _REGION_CHILD.format(root=ROOT), p] + specs,
                           env=dict(os.environ, FTK_DECODE_TIMING="1", **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]
        if os.environ.get("FTK_DEVICE_INFLATE") == "0":
            continue  # (the suite's inner run with the inflate on the host threads: regions are whole contigs there)
        assert "region 6000000-6100000" in r.stderr and "a long row hides the end, reading on" in r.stderr, r.stderr[-3000:]
        rep = {k: (a, b) for k, a, b in eval(r.stdout[r.stdout.index("ok") + 3:])}
        assert rep["c1:30000000:31000000"][0] < rep["c1:30000000:31000000"][1] // 20  # ~1.7 % of the contig's rows wanted
        assert rep["c2:10000000:10100000"][0] < rep["c2:10000000:10100000"][1] // 20
        assert rep["c3:0:5000000"][0] == rep["c3:0:5000000"][1]


def test_interval_calls_read_a_region_not_the_contig(tmp_path):
    """frag.wps / cleavage_profile / frag_length / single_coverage / frag_array on an interval of a contig that is not
    resident: the first two calls read the interval's rows through the index (a region table; the contig is NOT
    decoded), the third decodes the contig - and every answer equals the one computed on the whole contig."""
    from finaletoolkit_amd import frag, source
    from finaletoolkit_amd.utils import frag_array
    rng = np.random.default_rng(8)
    rows = []
    for name, n, size in (("c1", 500_000, 50_000_000), ("c2", 200_000, 20_000_000)):
        s = np.sort(rng.integers(0, size, n))
        rows.append((name, s, s + rng.integers(40, 500, n), rng.integers(0, 61, n), rng.integers(0, 2, n)))
    p = str(tmp_path / "iv.frag.gz")
    bgzf.write_frag_gz(p, rows, with_index=True)

    def calls():
        return dict(
            wps=frag.wps(p, "c1", 30_000_000, 30_020_000, 50_000_000)["wps"].tolist(),
            clv=frag.cleavage_profile(p, 50_000_000, "c1", 12_000_000, 12_003_000)["proportion"].tolist(),
            fl=frag.frag_length(p, "c2", 5_000_000, 5_400_000, intersect_policy="any").tolist(),
            cov=frag.single_coverage(p, "c2", 100_000, 900_000)[-1],
            arr=frag_array(p, "c1", 0, 49_990_000, 50_000_000, intersect_policy="any").tolist())

    source.close_all()
    del source.REGION_READS[:]
    first = calls()
    src = source.open_source(p)
    # c1: wps + cleavage as regions, the third call (frag_array) decoded the contig; c2: two region calls
    assert [r[1] for r in source.REGION_READS] == ["c1", "c1", "c2", "c2"], source.REGION_READS
    assert src.loaded == {"c1"} and len(src.regions) == 2
    source.close_all()
    for c in ("c1", "c2"):
        source.open_source(p).require(c)
    n_before = len(source.REGION_READS)
    whole = calls()
    assert len(source.REGION_READS) == n_before  # (resident contigs are used as they are)
    assert first == whole and len(first["wps"]) == 20_000 and any(first["wps"]) and first["cov"] > 2_000 and len(first["arr"]) > 50
    source.close_all()
