"""GPU: DEFLATE on the device (csrc/ftk_inflate.hip, one wavefront per BGZF block) against zlib on the host:
every block type (stored, fixed, dynamic), compressors and levels (zlib 0-9 with its strategies, libdeflate through
the library's own BGZF writer), data shapes (fragment rows, incompressible bytes, long runs, matches that reach
back further than the LDS window), block sizes from empty to 0xFF00, and damaged payloads (an error, not a hang)."""
import ctypes as C
import struct
import zlib

import numpy as np
import pytest

from finaletoolkit_amd import _lib as L, bgzf, synth, writers

pytestmark = pytest.mark.gpu


def _member(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, extra=b"") -> bytes:
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    payload = c.compress(data) + c.flush()
    xlen = 6 + len(extra)
    bsize = 12 + xlen + len(payload) + 8
    if bsize > 65536:
        return None  # does not fit a BGZF block (expanding strategy on incompressible data)
    head = struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, xlen) + extra + struct.pack("<BBHH", 66, 67, 2, bsize - 1)
    return head + payload + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data))


def _inflate_once(engine, image: bytes):
    lib = engine.lib
    n_out = C.c_int64()
    buf = np.zeros(1, np.uint8)
    rc = lib.ftk_bgzf_inflate_device(engine.ctx, image, len(image), L.ptr(buf), 0, C.byref(n_out))
    if rc == L.FTK_ERR_INVALID and n_out.value > 0:
        buf = np.zeros(n_out.value, np.uint8)
        rc = lib.ftk_bgzf_inflate_device(engine.ctx, image, len(image), L.ptr(buf), len(buf), C.byref(n_out))
    return rc, buf[:n_out.value].tobytes()


def _inflate(engine, image: bytes):
    """Every symbol loop of the kernel on the same image - the windowed loop with a window's matches copied one after
    the other (text streams) or resolved on the lanes side by side (BAM streams; ``FTK_INFLATE_VECTOR_MATCHES=1``), and
    the lane-parallel loop, the default at every launch size (``FTK_INFLATE_LANES`` unset or non-zero; ``=0`` selects the
    windowed loop) - must agree; the callers then hold the result against zlib."""
    import os
    names = ("FTK_INFLATE_VECTOR_MATCHES", "FTK_INFLATE_LANES")
    keep = {k: os.environ.get(k) for k in names}
    try:
        got = []
        for vec, lanes in (("0", "0"), ("1", "0"), ("0", "1")):
            os.environ.update(FTK_INFLATE_VECTOR_MATCHES=vec, FTK_INFLATE_LANES=lanes)
            got.append(_inflate_once(engine, image))
    finally:
        for k, v in keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    serial, vector, lanes = got
    assert serial[0] == vector[0] and (serial[0] != L.FTK_OK or serial[1] == vector[1])
    # (a damaged payload may trip a different check first in the lane-parallel loop: both must refuse it)
    assert (serial[0] == L.FTK_OK) == (lanes[0] == L.FTK_OK) and (serial[0] != L.FTK_OK or serial[1] == lanes[1])
    return vector


def _rows(n, seed):
    s, e, q, st = synth.synth_contig(max(2000, n * 10), 30.0, seed)
    return "".join(f"chr{seed}\t{a}\t{b}\t{m}\t{'+' if t else '-'}\n" for a, b, m, t in
                   zip(s[:n].tolist(), e[:n].tolist(), q[:n].tolist(), st[:n].tolist())).encode()


def test_block_types_levels_and_shapes(engine):
    rng = np.random.default_rng(1)
    text = _rows(2300, 3)[:0xFF00]
    noise = rng.integers(0, 256, 0xFF00, dtype=np.uint8).tobytes()
    far = rng.integers(0, 256, 30_000, dtype=np.uint8).tobytes()
    shapes = {
        "rows": text, "noise": noise, "zeros": bytes(0xFF00), "run_pattern": (b"abcdefg" * 10_000)[:0xFF00],
        "far_matches": (far + far)[:0xFF00],            # distance 30 000: beyond the LDS ring
        "mid_matches": (far[:7_000] + far[:7_000] * 8)[:0xFF00],  # distance 7 000..: ring edge
        # distances on both sides of the LDS-to-LDS limit (ring size - 322) and of the ring size, for a 2 KB and a 4 KB ring:
        # 258-byte matches whose sources straddle the line between "still in the ring" and "read back from HBM"
        **{f"edge_{d}": (far[:d] + far[:d] * 40)[:0xFF00] for d in (1_700, 1_726, 1_727, 1_800, 2_047, 2_048, 2_049, 2_300,
                                                                 3_774, 3_775, 4_095, 4_096, 4_097)},
        "edge_mixed": b"".join(far[k:k + 300] for k in (0, 1_400, 0, 1_700, 0, 2_000, 1_400, 2_100, 0)) * 12,
        "one_byte": b"x", "empty": b"", "short": b"hello, hello, hello\n",
        "digits": "".join(str(v) for v in rng.integers(0, 10 ** 9, 7000)).encode()[:0xFF00],
    }
    members, want = [], []
    for name, data in shapes.items():
        for level, strategy in [(0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY),
                                (9, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE),
                                (6, zlib.Z_FILTERED)]:
            m = _member(data, level, strategy)
            if m is not None:
                members.append(m)
                want.append(data)
    members.append(_member(text[:5000], 6, extra=struct.pack("<BBH", 88, 89, 3) + b"abc"))  # another extra subfield first
    want.append(text[:5000])
    image = b"".join(members) + bgzf._EOF
    rc, got = _inflate(engine, image)
    assert rc == 0, engine.lib.ftk_last_error(engine.ctx)
    assert got == b"".join(want)
    # every member alone too (an error names the block)
    for m, w in zip(members[:12], want[:12]):
        rc, got = _inflate(engine, m)
        assert rc == 0 and got == w


def test_inflate_fuzz_random_mixtures(engine):
    """300 blocks of random make-up - stretches of random bytes, of a few symbols, of one byte, and copies of earlier
    stretches from random distances (1 byte to 40 KB back: inside the LDS window, at its edge, beyond it) - through
    random zlib levels and strategies: byte-equal to the input, block after block."""
    rng = np.random.default_rng(77)
    members, want = [], []
    while len(members) < 300:
        out = bytearray()
        target = int(rng.integers(1, 0xFF00))
        while len(out) < target:
            kind = int(rng.integers(0, 5))
            n = int(rng.integers(1, 3000))
            if kind == 0:
                out += rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            elif kind == 1:
                out += rng.integers(48, 58, n, dtype=np.uint8).tobytes()
            elif kind == 2:
                out += bytes([int(rng.integers(0, 256))]) * n
            elif out:
                d = int(min(len(out), rng.choice([1, 2, 3, 7, 64, 258, 700, 1_700, 1_726, 1_727, 2_048, 2_049, 3_775, 4_096,
                                                   4_097, 9_000, 32_768, 40_000])))
                for _ in range(n // d + 1):  # (overlapping copies when n > d)
                    out += out[len(out) - d:len(out) - d + min(d, n)]
        data = bytes(out[:target])
        level = int(rng.integers(0, 10))
        strategy = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY][int(rng.integers(0, 5))]
        m = _member(data, level, strategy)
        if m is not None:
            members.append(m)
            want.append(data)
    rc, got = _inflate(engine, b"".join(members) + bgzf._EOF)
    assert rc == 0, engine.lib.ftk_last_error(engine.ctx)
    expect = b"".join(want)
    if got != expect:  # name the first block that differs
        off = 0
        for k, w in enumerate(want):
            assert got[off:off + len(w)] == w, f"block {k} ({len(w)} bytes) differs"
            off += len(w)
    assert got == expect


def test_windowed_symbol_loop_edges(engine):
    """Payloads aimed at the windowed symbol loop: alphabets skewed so that rare literals and rare lengths get codes longer
    than the 9-bit root (a window stops in front of them, the serial path takes one symbol); windows whose matches add up to
    more than the 320 bytes a window may produce (258-byte matches back to back, cut between symbols); many short matches
    that read what the previous match of the same window wrote (distance 1-8 behind a literal); matches with every length
    3-258 at distances on both sides of the window's start; members made of many small DEFLATE blocks (Z_FULL_FLUSH
    every few hundred bytes: new tables in the middle of a window's worth of bits); a block that ends exactly where a
    window would."""
    rng = np.random.default_rng(5)
    shapes = {}
    # 1. geometric alphabet: symbol k with probability ~ 2^-k/3 over 200 symbols -> code lengths 1..15
    p = 0.79 ** np.arange(200)
    shapes["skewed"] = rng.choice(200, 60_000, p=p / p.sum()).astype(np.uint8).tobytes()
    # ... and the same with stretches copied from a little earlier (rare lengths, rare distances)
    sk = bytearray(shapes["skewed"][:20_000])
    for _ in range(900):
        a = int(rng.integers(0, len(sk) - 300))
        n = int(rng.choice([3, 4, 5, 9, 17, 33, 65, 129, 200, 257, 258]))
        sk += sk[a:a + n] + bytes(rng.integers(0, 200, int(rng.integers(0, 4)), dtype=np.uint8))
    shapes["skewed_matches"] = bytes(sk[:0xFF00])
    # 2. back-to-back long matches
    unit = rng.integers(0, 256, 300, dtype=np.uint8).tobytes()
    shapes["long_matches"] = (unit * 220)[:0xFF00]
    # 3. short dependent matches: x, then x repeated (distance 1), then a 2-byte pattern, ... separated by fresh literals
    dep = bytearray()
    while len(dep) < 60_000:
        k = int(rng.integers(1, 9))
        pat = rng.integers(0, 256, k, dtype=np.uint8).tobytes()
        dep += pat * int(rng.integers(2, 12)) + rng.integers(0, 256, int(rng.integers(1, 5)), dtype=np.uint8).tobytes()
    shapes["dependent_matches"] = bytes(dep[:60_000])
    # 4. every match length at assorted distances
    base = rng.integers(0, 256, 4_000, dtype=np.uint8).tobytes()
    ev = bytearray(base)
    for n in range(3, 259):
        d = int(rng.choice([n, n + 1, 2 * n, 300, 1_000, 1_726, 1_727, 3_000]))
        d = min(d, len(ev))
        ev += ev[len(ev) - d:len(ev) - d + n] if d >= n else (ev[len(ev) - d:] * (n // d + 1))[:n]
        ev += bytes([n & 255, 255 - (n & 255)])
    shapes["every_length"] = bytes(ev[:0xFF00])
    members, want = [], []
    for name, data in shapes.items():
        for level, strategy in [(6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY),
                                (6, zlib.Z_FILTERED), (6, zlib.Z_HUFFMAN_ONLY)]:
            m = _member(data, level, strategy)
            if m is not None:
                members.append(m)
                want.append(data)
    # 5. many small DEFLATE blocks inside one member
    text = _rows(2000, 9)[:50_000]
    for step in (97, 256, 1000):
        c = zlib.compressobj(6, zlib.DEFLATED, -15)
        payload = b"".join(c.compress(text[i:i + step]) + c.flush(zlib.Z_FULL_FLUSH if (i // step) % 3 else zlib.Z_SYNC_FLUSH)
                           for i in range(0, len(text), step)) + c.flush()
        bsize = 12 + 6 + len(payload) + 8
        if bsize <= 65536:
            head = struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + struct.pack("<BBHH", 66, 67, 2, bsize - 1)
            members.append(head + payload + struct.pack("<II", zlib.crc32(text) & 0xFFFFFFFF, len(text)))
            want.append(text)
    assert len(members) >= 20
    rc, got = _inflate(engine, b"".join(members) + bgzf._EOF)
    assert rc == 0, engine.lib.ftk_last_error(engine.ctx)
    off = 0
    for k, w in enumerate(want):
        assert got[off:off + len(w)] == w, f"member {k} ({len(w)} bytes) differs"
        off += len(w)
    assert off == len(got)


def test_many_blocks_from_the_library_writer(engine, tmp_path):
    """A 40 MB fragment text through the library's BGZF writer (libdeflate) at three levels: ~600 blocks each."""
    rows = _rows(1_300_000, 5)
    for level in (1, 6, 12):
        p = tmp_path / f"l{level}.gz"
        writers.bgzf_write(str(p), rows, level)
        image = open(p, "rb").read()
        rc, got = _inflate(engine, image)
        assert rc == 0, engine.lib.ftk_last_error(engine.ctx)
        assert got == rows, level


def test_image_beyond_one_launch_of_the_kernel(engine, tmp_path):
    """The kernel addresses its input by 32-bit bit positions: ``ftk_bgzf_inflate_device`` cuts an image whose compressed
    bytes span more than 2^28 into launches from rebased pointers (until round 5 the blocks behind 2^29 bytes did not
    decode: 'reason 5').  ~340 MB of BAM records -> 750 MB, both symbol loops, against zlib."""
    import gzip
    import os
    p = str(tmp_path / "big.bam")
    synth.write_paired_bam_native(p, [("x", 14_000_000)], 60.0, 31, keep=())
    image = open(p, "rb").read()
    assert len(image) > (1 << 28) + (1 << 24)
    text = gzip.open(p, "rb").read()
    keep = os.environ.get("FTK_INFLATE_LANES")
    try:
        for lanes in ("1", "0"):
            os.environ["FTK_INFLATE_LANES"] = lanes
            out = np.zeros(len(text), np.uint8)
            n = C.c_int64()
            rc = engine.lib.ftk_bgzf_inflate_device(engine.ctx, image, len(image), L.ptr(out), len(out), C.byref(n))
            assert rc == L.FTK_OK, engine.lib.ftk_last_error(engine.ctx)
            assert n.value == len(text) and out.tobytes() == text
    finally:
        if keep is None:
            os.environ.pop("FTK_INFLATE_LANES", None)
        else:
            os.environ["FTK_INFLATE_LANES"] = keep


def test_damaged_payloads_are_errors(engine):
    text = _rows(2000, 9)
    good = _member(text, 6)
    rng = np.random.default_rng(2)
    bad = 0
    for k in range(40):
        m = bytearray(good)
        pos = int(rng.integers(18, len(m) - 8))
        m[pos] ^= 1 << int(rng.integers(0, 8))
        rc, got = _inflate(engine, bytes(m))
        # a flipped bit may well decode to the right number of bytes (a different literal): the CRC catches those
        assert rc == L.FTK_ERR_FORMAT, (k, pos)
        bad += 1
    assert bad == 40
    m = bytearray(good)
    m[-8] ^= 1  # the trailer's CRC itself
    assert _inflate(engine, bytes(m))[0] == L.FTK_ERR_FORMAT
    # wrong ISIZE, truncated payload, reserved block type
    m = bytearray(good)
    m[-4:] = struct.pack("<I", len(text) + 5)
    assert _inflate(engine, bytes(m))[0] == L.FTK_ERR_FORMAT
    m = bytearray(good)
    m[-4:] = struct.pack("<I", len(text) - 5)
    assert _inflate(engine, bytes(m))[0] == L.FTK_ERR_FORMAT
    reserved = bytearray(_member(b"abc", 6))
    reserved[18] |= 0x06  # BTYPE = 3
    assert _inflate(engine, bytes(reserved))[0] == L.FTK_ERR_FORMAT
    assert _inflate(engine, b"\x1f\x8b\x08\x00" + bytes(30))[0] == L.FTK_ERR_FORMAT  # gzip without the BGZF field


@pytest.mark.parametrize("module", ["tests/test_gpu_device_parse.py", "tests/test_gpu_api_golden.py tests/test_gpu_cli.py"])
def test_decoder_suites_with_host_inflate(module):
    """The streaming decoder inflates BGZF blocks on the GPU by default (carry and line ends found there too: the host
    never sees the text) -- that is what every other test of the suite runs.  FTK_DEVICE_INFLATE=0 keeps the inflate
    on the host threads: the decoder's own test suites -- fixtures, BED6, CRLF, no final newline, contig runs inside
    pieces, host-parser fall-backs, index-driven single contigs, truncated / unsorted files, and the
    reference-shaped API goldens on top -- must pass that way too."""
    import os
    import subprocess
    import sys
    if os.environ.get("FTK_DEVICE_INFLATE") == "0":
        pytest.skip("this is the inner run")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", *module.split()], cwd=root,
                       env=dict(os.environ, FTK_DEVICE_INFLATE="0"), capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


_BAM_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
from finaletoolkit_amd import source, synth
from oracle import oracle as O
size = {size}
exp = synth.write_paired_bam({bam!r}, "mid", size, 60.0, 17)
names = []
for src, name in source.stream_source({bam!r}):
    names.append(name)
assert names == ["mid"], names
eng = source.get_engine()
key = src.key("mid")
assert eng.info(key)[0] == exp["n"], (eng.info(key)[0], exp["n"])
fr = O.Frags(exp["s"], exp["e"], exp["q"], exp["st"], exp["r1s"], exp["r1e"])
ws, we = synth.tiling_windows(size, 20_000)
f = eng.window_features(key, ws, we, 30, hist=(0, 1001))
h, o = O.c_fraglen_hist(fr, ws, we, 0, 1001, mapq_min=30)
assert np.array_equal(f["coverage"], O.c_window_counts(fr, ws, we, mapq_min=30))
assert np.array_equal(f["hist"], h) and np.array_equal(f["overflow"], o)
a = size // 3
assert np.array_equal(eng.wps(key, a, a + 50_000, size), O.c_wps(fr, a, a + 50_000, size))
print("ok", exp["n"])
"""


@pytest.mark.parametrize("piece", [1 << 17, 3 << 20])
def test_bam_stream_many_pieces_through_the_slot_ring(tmp_path, piece):
    """A BAM stream inflates its pieces on the device with several pieces in flight (run_bam: a ring of slots, one
    HIP stream each).  Small pieces make one file go round the ring dozens of times - records cut by piece ends
    (carried on the host), the header in the first piece, speculative stretches of the record chain - and the
    fragments must be the ones the synthesizer wrote; with the inflate on the host threads likewise."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _BAM_CHILD.format(root=root, size=600_000, bam=str(tmp_path / "ring.bam"))
    outs = []
    # records parsed on the device (the default) / inflate on the device, records walked by the host / all on the host
    for dinf, drec in (("1", "1"), ("1", "0"), ("0", "0")):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                           env=dict(os.environ, FTK_STREAM_PIECE=str(piece), FTK_BAM_STRETCH="65536", FTK_DEVICE_INFLATE=dinf,
                                    FTK_DEVICE_BAM_PARSE=drec, FTK_BAM_DEV_STRETCH="2048"))
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] == outs[2] and outs[0].startswith("ok")


_BAM_MULTI_CHILD = r"""
import sys, ctypes as C, numpy as np
sys.path.insert(0, {root!r})
import torch  # noqa: F401  (one HIP runtime in the process)
from finaletoolkit_amd import _lib as L
from tests.test_stream_decoder import _same
from tests.test_abi import _decode
lib = L.load()
path = sys.argv[1]
n_device = [0]


def stream(contig=None, threads=4):
    s = C.c_void_p()
    rc = lib.ftk_fragstream_open_device(0, path.encode(), None if contig is None else contig.encode(), 1, threads, 2, C.byref(s))
    assert rc == 0, lib.ftk_fragtable_error().decode()
    out, order = {{}}, []
    while True:
        t = C.c_void_p()
        rc = lib.ftk_fragstream_next(s, C.byref(t))
        assert rc == 0, lib.ftk_fragtable_error().decode()
        if not t.value:
            break
        rows = lib.ftk_fragtable_contig_rows(t, 0)
        name = lib.ftk_fragtable_contig_name(t, 0).decode()
        if lib.ftk_fragtable_is_device(t, 0):  # records parsed on the device: the columns live in HBM
            cols = [np.empty(rows, dt) for dt in (np.int32, np.int32, np.uint8, np.uint8, np.int32, np.int32)]
            assert lib.ftk_fragtable_columns_to_host(t, 0, *[c.ctypes.data_as(C.c_void_p) for c in cols[:4]]) == 0
            rank = np.empty(rows, np.int32)
            assert lib.ftk_fragtable_read1_to_host(t, 0, cols[4].ctypes.data_as(C.c_void_p), cols[5].ctypes.data_as(C.c_void_p),
                                                   rank.ctypes.data_as(C.c_void_p)) == 0
            # the file-order rank is a permutation and the rows are in stable start order
            assert rows == 0 or (np.array_equal(np.sort(rank), np.arange(rows)) and np.all(np.diff(cols[0]) >= 0))
            n_device[0] += 1
        else:
            ps = [C.c_void_p() for _ in range(6)]
            assert lib.ftk_fragtable_columns(t, 0, *[C.byref(p) for p in ps]) == 0
            cols = [None if not p.value or rows == 0 else np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), (rows,)).copy()
                    for p, ct in zip(ps, (C.c_int32, C.c_int32, C.c_uint8, C.c_uint8, C.c_int32, C.c_int32))]
        out[name] = (rows, cols, lib.ftk_fragtable_contig_length(t, 0))
        order.append(name)
        lib.ftk_fragtable_free(t)
    lib.ftk_fragstream_close(s)
    return out, order


want = _decode(path, bam=True)
for threads in (1, 8):
    got, order = stream(threads=threads)
    _same(got, want)
only, order1 = stream("chrB")
assert order1 == ["chrB"], order1
_same(only, {{k: v for k, v in want.items() if k in ("chrB",) or k.startswith("__")}})
print("ok", order, "device_tables", n_device[0])
"""


@pytest.mark.parametrize("piece,stretch,ramp", [(1 << 16, "64", None), (1 << 16, "700", None), (1 << 18, "4096", None),
                                                (1 << 20, "16384", None), (48 << 20, "16384", None),
                                                (1 << 20, "4096", "65536"), (1 << 19, "16384", "32768")])
def test_bam_records_parsed_on_the_device(tmp_path, piece, stretch, ramp):
    """The same file with the RECORDS PARSED ON THE DEVICE (run_bam_device, ftk_bamparse.hip; the default): stretches
    from 64 bytes (several per record: most guesses are wrong and the fix passes settle the chain) to 16 KB, pieces from
    64 KB (records and the header's tail cut by piece ends, contig changes inside pieces) to one piece for the file; the
    tables hold device columns in stable fragment-start order and equal the whole-file host decoder's; one contig
    through the BAI.  ``ramp``: the stream's first reads short, doubling up to the piece size (FTK_STREAM_RAMP)."""
    import os
    import subprocess
    import sys
    from tests.helpers import write_synthetic_bam
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(29)
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
    env = dict(os.environ, FTK_STREAM_PIECE=str(piece), FTK_BAM_DEV_STRETCH=stretch, FTK_DECODE_TIMING="1")
    if ramp:
        env["FTK_STREAM_RAMP"] = ramp
    r = subprocess.run([sys.executable, "-c", _BAM_MULTI_CHILD.format(root=root), p], capture_output=True, text=True, timeout=900,
                       env=env)
    assert r.returncode == 0 and "ok ['chrA', 'chrB', 'chrC']" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]
    assert "device_tables 7" in r.stdout, r.stdout[-500:]  # 3 + 3 + 1 tables, none through the host fall-back
    assert "parsed on the device" in r.stderr and "stretches of the record chain" not in r.stderr, r.stderr[-1500:]


@pytest.mark.parametrize("piece,host_share", [(1 << 16, "3"), (1 << 16, "0"), (1 << 16, "1"), (1 << 18, "2")])
def test_bam_multi_contig_through_the_slot_ring(tmp_path, piece, host_share):
    """Three contigs with fragments and one without, unmapped / secondary / duplicate records in between
    (tests/helpers.write_synthetic_bam), read in 64 KB pieces through ftk_fragstream_open_device: contig changes inside
    pieces, records cut by piece ends, the BAI seek for one contig, every share of the pieces inflated by the host
    threads beside the GPU (FTK_BAM_HOST_SHARE 0 / 1 / 2 / 3) - the tables are those of the whole-file host decoder."""
    import os
    import subprocess
    import sys
    from tests.helpers import write_synthetic_bam
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(19)
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
    r = subprocess.run([sys.executable, "-c", _BAM_MULTI_CHILD.format(root=root), p], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, FTK_STREAM_PIECE=str(piece), FTK_BAM_STRETCH="4096", FTK_BAM_HOST_SHARE=host_share,
                                FTK_DEVICE_BAM_PARSE="0"))
    assert r.returncode == 0 and "ok ['chrA', 'chrB', 'chrC']" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]


_BAM_DAMAGED_CHILD = r"""
import sys, ctypes as C
sys.path.insert(0, {root!r})
import torch  # noqa: F401
from finaletoolkit_amd import _lib as L
lib = L.load()
s = C.c_void_p()
rc = lib.ftk_fragstream_open_device(0, sys.argv[1].encode(), None, 1, 4, 2, C.byref(s))
assert rc == 0
n = 0
while True:
    t = C.c_void_p()
    rc = lib.ftk_fragstream_next(s, C.byref(t))
    if rc != 0:
        print("error", rc, lib.ftk_fragtable_error().decode())
        break
    if not t.value:
        print("complete", n)
        break
    n += 1
    lib.ftk_fragtable_free(t)
lib.ftk_fragstream_close(s)
"""


@pytest.mark.parametrize("host_share,device_records", [("0", "0"), ("1", "0"), ("3", "0"), ("0", "1")])
def test_bam_stream_with_a_damaged_block_is_an_error(tmp_path, host_share, device_records):
    """One flipped payload byte somewhere in the middle of a BAM: whichever side inflates that piece (the GPU, checked
    by its CRC kernel, or the host threads beside it), the stream ends with a format error - no crash, no hang, no
    silently different fragments."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    good = str(tmp_path / "good.bam")
    synth.write_paired_bam(good, "mid", 400_000, 60.0, 23)
    image = bytearray(open(good, "rb").read())
    # the payload of the block that holds the middle of the file: flip a byte well inside it
    off, blocks = 0, []
    while off < len(image):
        bs = int.from_bytes(image[off + 16:off + 18], "little") + 1
        blocks.append((off, bs))
        off += bs
    o, bs = blocks[len(blocks) // 2]
    image[o + 18 + (bs - 26) // 2] ^= 0x5A
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(bytes(image))
    code = _BAM_DAMAGED_CHILD.format(root=root)
    env = dict(os.environ, FTK_STREAM_PIECE=str(1 << 17), FTK_BAM_HOST_SHARE=host_share, FTK_DEVICE_BAM_PARSE=device_records)
    r = subprocess.run([sys.executable, "-c", code, good], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "complete 1" in r.stdout, r.stdout + r.stderr[-1500:]
    r = subprocess.run([sys.executable, "-c", code, bad], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.startswith("error"), r.stdout + r.stderr[-1500:]
    assert str(L.FTK_ERR_FORMAT) in r.stdout.split()[1]


def test_cache_trim_releases_the_streams_idle_buffers(tmp_path):
    """After a stream the library holds its buffer sets, page-locked tables and result blocks for the next one;
    ftk_cache_trim gives them back (bytes > 0), a second trim finds nothing, and the next stream works as before."""
    from finaletoolkit_amd import source
    size = 3_000_000
    s, e, q, st = synth.synth_contig(size, 30.0, 77)
    p = str(tmp_path / "t.frag.gz")
    bgzf.write_frag_gz(p, [("c", s, e, q, st)], level=1, with_index=False)
    ws, we = synth.tiling_windows(size, 100_000)

    def run():
        source.close_all()
        eng = source.get_engine()
        for src, name in source.stream_source(p):
            r = eng.window_features(src.key(name), ws, we, 30)
            w = eng.wps(src.key(name), 0, size, size)
        cov = int(r["coverage"].sum())
        del r, w
        source.close_all()
        return cov

    a = run()
    freed = source.release_caches()
    assert freed > 0
    assert source.release_caches() == 0
    assert run() == a


def _multi_child(tmp_path, bam_path, want_order, env):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    child = _BAM_MULTI_CHILD.replace('only, order1 = stream("chrB")', 'only, order1 = stream(%r)' % want_order[1]) \
                            .replace('assert order1 == ["chrB"], order1', 'assert order1 == [%r], order1' % want_order[1]) \
                            .replace('if k in ("chrB",)', 'if k in (%r,)' % want_order[1])
    r = subprocess.run([sys.executable, "-c", child.format(root=root), bam_path], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, FTK_DECODE_TIMING="1", **env))
    assert r.returncode == 0 and r.stdout.startswith("ok"), r.stdout[-1500:] + r.stderr[-2500:]
    return r


def test_bam_device_parser_hands_over_to_the_host_decoder(tmp_path):
    """More contig runs in one piece than the device summary lists (a BAM's decoy / alt contigs: 150 contigs with a
    handful of reads each, all in one 64 KB piece) - the stream starts over on the host decoder, which skips the
    contigs the device pass had already handed out; the consumer sees every contig once, with the whole-file
    decoder's rows."""
    from tests.helpers import write_synthetic_bam
    rng = np.random.default_rng(41)
    contigs = [("big0", 600_000)] + [(f"alt{k:03d}", 20_000) for k in range(150)] + [("big1", 300_000)]
    frags = {}
    for name, size in contigs:
        n = size // 50 if name.startswith("big") else 6
        s = np.sort(rng.integers(0, size - 700, n))
        frags[name] = (s, s + rng.integers(210, 600, n), rng.integers(0, 61, n), rng.integers(0, 2, n).astype(bool))
    p = str(tmp_path / "alts.bam")
    write_synthetic_bam(p, contigs, frags)
    r = _multi_child(tmp_path, p, ("big0", "alt077"), dict(FTK_STREAM_PIECE=str(1 << 16)))
    assert "the host decoder takes over" in r.stderr, r.stderr[-1500:]
    assert "'big0', 'alt000'" in r.stdout and "'alt149', 'big1']" in r.stdout, r.stdout[-600:]
    # with pieces that hold fewer runs than the limit the device parses the whole file
    r = _multi_child(tmp_path, p, ("big0", "alt077"), dict(FTK_STREAM_PIECE=str(1 << 16), FTK_BAM_DEV_STRETCH="512"))
    assert "'alt149', 'big1']" in r.stdout


def test_bam_whose_compression_rises_behind_its_first_piece_stays_on_the_device(tmp_path):
    """A large BAM is read in doubled pieces, sized for records that deflate 3-5 x.  A file whose later records deflate
    far better can outgrow the text a piece may hold (4 GiB; FTK_TEST_PIECE_LIMIT shrinks that limit and applies the
    doubling to a small file): the stream then starts over ON THE DEVICE with standard pieces - not on the host decoder -,
    skips the contig it had handed out, and every contig arrives once with the whole-file decoder's rows."""
    import gzip
    import os
    from tests.helpers import write_synthetic_bam
    rng = np.random.default_rng(47)
    contigs = [("first", 3_600_000), ("second", 4_000_000)]
    frags = {}
    for name, size in contigs:
        n = size // 30
        s = np.sort(rng.integers(0, size - 2_000, n))
        frags[name] = (s, s + rng.integers(420, 900, n), rng.integers(0, 61, n), rng.integers(0, 2, n).astype(bool))
    # two files with one header, spliced: short reads (names dominate: ~5 x) in front, long constant reads (~40 x) behind
    a, b = str(tmp_path / "a.bam"), str(tmp_path / "b.bam")
    write_synthetic_bam(a, contigs, {"first": frags["first"]}, read_len=30, junk=False)
    write_synthetic_bam(b, contigs, {"second": frags["second"]}, read_len=400, junk=False)
    ra, rb = gzip.open(a, "rb").read(), gzip.open(b, "rb").read()
    head = 12 + int.from_bytes(ra[4:8], "little")
    head += sum(8 + len(c) + 1 for c, _ in contigs)
    assert ra[:head] == rb[:head]
    from finaletoolkit_amd import bgzf
    p = str(tmp_path / "rising.bam")
    offs = bgzf.write_bgzf(p, ra + rb[head:], level=6)
    bgzf.write_index(p + ".bai", True, [("first", bgzf.virtual_offset(offs, head), bgzf.virtual_offset(offs, len(ra))),
                                        ("second", bgzf.virtual_offset(offs, len(ra)), bgzf.virtual_offset(offs, len(ra) + len(rb) - head))])
    ratio_first = len(ra) / os.path.getsize(a)
    ratio_second = (len(rb) - head) / max(os.path.getsize(p) - os.path.getsize(a), 1)
    assert ratio_first < 14 and ratio_second > 1.6 * ratio_first, (ratio_first, ratio_second)
    piece = 1 << 20
    room = 32 << 20  # (kRoom of the stream: a piece's text is held behind room for the carried record)
    limit = room + int(1.5 * piece * ratio_second)  # a standard piece of the second contig fits, a doubled one does not
    assert room + 2 * piece * ratio_first < limit and os.path.getsize(a) > 2.5 * piece and os.path.getsize(p) - os.path.getsize(a) > 3 * piece
    r = _multi_child(tmp_path, p, ("first", "second"), dict(FTK_STREAM_PIECE=str(piece), FTK_TEST_PIECE_LIMIT=str(limit)))
    assert "the device path starts over with standard pieces" in r.stderr and "the host decoder takes over" not in r.stderr, r.stderr[-2000:]
    assert "['first', 'second']" in r.stdout, r.stdout[-600:]


def test_bam_device_parser_with_records_longer_than_a_stretch(tmp_path):
    """Long reads: records of 6-8 KB (read length 4 000) against stretches of 1 KB and 16 KB - most stretches hold no
    record start at all, the guesses land inside sequence bytes, and the chain still settles on the device."""
    from tests.helpers import write_synthetic_bam
    rng = np.random.default_rng(43)
    contigs = [("chrL", 4_000_000), ("chrM", 900_000)]
    frags = {}
    for name, size in contigs:
        n = size // 2_000
        s = np.sort(rng.integers(0, size - 12_000, n))
        frags[name] = (s, s + rng.integers(4_100, 9_000, n), rng.integers(0, 61, n), rng.integers(0, 2, n).astype(bool))
    p = str(tmp_path / "long.bam")
    write_synthetic_bam(p, contigs, frags, read_len=4_000)
    for stretch in ("1024", "16384"):
        r = _multi_child(tmp_path, p, ("chrL", "chrM"), dict(FTK_STREAM_PIECE=str(1 << 18), FTK_BAM_DEV_STRETCH=stretch))
        assert "['chrL', 'chrM']" in r.stdout and "the host decoder takes over" not in r.stderr, r.stdout[-400:] + r.stderr[-1500:]


_BAM_REGION_CHILD = """
import ctypes as C, sys
import numpy as np
sys.path.insert(0, {root!r})
from finaletoolkit_amd import _lib as L
lib = L.load()

def table_rows(t):
    rows = lib.ftk_fragtable_contig_rows(t, 0)
    cols = [np.empty(rows, np.int32), np.empty(rows, np.int32), np.empty(rows, np.uint8), np.empty(rows, np.uint8)]
    assert lib.ftk_fragtable_columns_to_host(t, 0, *[c.ctypes.data_as(C.c_void_p) for c in cols]) == 0
    r1 = [np.empty(rows, np.int32), np.empty(rows, np.int32)]
    assert lib.ftk_fragtable_read1_to_host(t, 0, *[c.ctypes.data_as(C.c_void_p) for c in r1], None) == 0
    return np.stack([c.astype(np.int64) for c in cols + r1], 1)

def stream(path, contig, a=None, b=None):
    s = C.c_void_p()
    if a is None:
        rc = lib.ftk_fragstream_open_device(0, path.encode(), contig.encode(), 1, 8, 1, C.byref(s))
    else:
        rc = lib.ftk_fragstream_open_region(0, path.encode(), contig.encode(), a, b, 1, 8, 1, C.byref(s))
    assert rc == 0, lib.ftk_fragtable_error().decode()
    out = None
    while True:
        t = C.c_void_p()
        rc = lib.ftk_fragstream_next(s, C.byref(t))
        assert rc == 0, lib.ftk_fragtable_error().decode()
        if not t.value:
            break
        assert out is None
        out = table_rows(t)
        lib.ftk_fragtable_free(t)
    lib.ftk_fragstream_close(s)
    return np.zeros((0, 6), np.int64) if out is None else out

path = sys.argv[1]
from collections import Counter
whole = stream(path, "mid")
keys = Counter(map(tuple, whole.tolist()))
report = []
for spec in sys.argv[2:]:
    a, b = map(int, spec.split(":"))
    got = stream(path, "mid", a, b)
    gk = Counter(map(tuple, got.tolist()))
    assert not (gk - keys), spec                                          # rows of the contig, none more often than there
    need = whole[(whole[:, 4] < b) & (whole[:, 5] > a)]                   # read1 overlaps the region
    assert not (Counter(map(tuple, need.tolist())) - gk), (spec, len(need), len(got))
    assert np.all(np.diff(got[:, 0]) >= 0), spec                          # sorted by fragment start like the whole table
    report.append((spec, len(got), len(whole), len(need)))
print("ok", report)
"""


def test_bam_region_streams_hold_every_read1_that_overlaps(tmp_path):
    """ftk_fragstream_open_region on a BAM with a linear index in its BAI: the records are read from the first that
    overlaps the region to the first behind it (the device parser's summary carries the last record's position), the
    table holds every fragment whose read1 overlaps the region, sorted by fragment start, and a small region reads a
    small part of the contig - with 48 MB and 128 KB pieces."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = str(tmp_path / "reg.bam")
    synth.write_paired_bam(p, "mid", 6_000_000, 30.0, 17)
    specs = ["0:50000", "3000000:3100000", "5950000:6000000", "2500000:2500001", "1000000:4000000", "5999999:7000000"]
    for env in (dict(), dict(FTK_STREAM_PIECE=str(1 << 17))):
        r = subprocess.run([sys.executable, "-c", _BAM_REGION_CHILD.format(root=root), p] + specs, capture_output=True, text=True,
                           timeout=900, env=dict(os.environ, **env))
        assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
        rep = {k: (a, b, c) for k, a, b, c in eval(r.stdout[r.stdout.index("ok") + 3:])}
        assert rep["3000000:3100000"][0] < rep["3000000:3100000"][1] // 10 and rep["3000000:3100000"][2] > 1000
        assert rep["1000000:4000000"][0] > rep["1000000:4000000"][1] // 2 - 1000
