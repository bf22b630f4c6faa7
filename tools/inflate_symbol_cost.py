#!/usr/bin/env python3
"""What a literal and a match cost the device inflate's decode chain: per-block wavefront lifetimes (library built with
-DFTK_INFLATE_TIMING, see tools/inflate_block_times.py) of 64 blocks - far fewer than the chip holds, so a block's
lifetime is its chain's latency - for payloads of one kind each."""
import ctypes as C
import os
import struct
import sys
import zlib

import numpy as np

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from finaletoolkit_amd import synth, _lib as L  # noqa: E402
from finaletoolkit_amd.engine import Engine  # noqa: E402


def member(data, level=6, strategy=zlib.Z_DEFAULT_STRATEGY):
    c = zlib.compressobj(level, zlib.DEFLATED, -15, 9, strategy)
    payload = c.compress(data) + c.flush()
    bsize = 12 + 6 + len(payload) + 8
    assert bsize <= 65536
    head = struct.pack("<BBBBIBBH", 31, 139, 8, 4, 0, 0, 255, 6) + struct.pack("<BBHH", 66, 67, 2, bsize - 1)
    return head + payload + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)), len(payload)


eng = Engine(0)
eng.lib.ftk_debug_inflate_ticks.argtypes = [C.c_void_p, C.c_int]
rng = np.random.default_rng(3)
n = 48_000
s, e, q, st = synth.synth_contig(20_000_000, 30.0, 5)
rows = "".join(f"20\t{a}\t{b}\t{m}\t{'+' if t else '-'}\n" for a, b, m, t in zip(s[:4000].tolist(), e[:4000].tolist(), q[:4000].tolist(), st[:4000].tolist())).encode()[:n]
cases = {
    "16 byte values, Huffman only (4-bit codes: two literals per hop)": (rng.choice(np.array([0x11, 0x12, 0x14, 0x18, 0x21, 0x22, 0x24, 0x28, 0x41, 0x42, 0x44, 0x48, 0x81, 0x82, 0x84, 0x88], np.uint8), n).tobytes(), zlib.Z_HUFFMAN_ONLY),
    "40 byte values, Huffman only (5-6-bit codes: one literal per hop)": (rng.integers(33, 73, n, dtype=np.uint8).tobytes(), zlib.Z_HUFFMAN_ONLY),
    "256 byte values, Huffman only (8-bit codes)": (rng.integers(0, 256, n, dtype=np.uint8).tobytes(), zlib.Z_HUFFMAN_ONLY),
    "fragment rows, level 6 (matches and literals)": (rows, zlib.Z_DEFAULT_STRATEGY),
    "one row repeated, level 6 (all matches of 258)": ((rows[:27] * (n // 27 + 1))[:n], zlib.Z_DEFAULT_STRATEGY),
}
for label, (data, strategy) in cases.items():
    m, plen = member(data, 6, strategy)
    image = m * 64
    out = np.zeros(len(data) * 64, np.uint8)
    got = C.c_int64()
    if hasattr(eng.lib, "ftk_debug_inflate_profile"):
        eng.lib.ftk_debug_inflate_profile.argtypes = [C.c_void_p, C.c_int]
        eng.lib.ftk_debug_inflate_profile(np.zeros(16, np.uint64).ctypes.data, 1)
    for _ in range(2):
        rc = eng.lib.ftk_bgzf_inflate_device(eng.ctx, image, len(image), L.ptr(out), len(out), C.byref(got))
        assert rc == 0, eng.lib.ftk_last_error(eng.ctx)
    assert out[:len(data)].tobytes() == data
    ticks = np.zeros(128, np.uint64)
    assert eng.lib.ftk_debug_inflate_ticks(ticks.ctypes.data, 64) == 0
    t = ticks.reshape(-1, 2).astype(np.int64)
    us = np.median(t[:, 1] - t[:, 0]) / 100.0
    if hasattr(eng.lib, "ftk_debug_inflate_profile"):
        prof = np.zeros(16, np.uint64)
        eng.lib.ftk_debug_inflate_profile.argtypes = [C.c_void_p, C.c_int]
        assert eng.lib.ftk_debug_inflate_profile(prof.ctypes.data, 1) == 0
        names = ["refill", "gathers+decode", "chain", "scan+stores", "match copies", "flush+drop", "window not taken", "other"]
        tot = float(prof[:8].sum())
        print("   " + "; ".join(f"{nm} {100 * float(prof[i]) / tot:.0f}% ({float(prof[i]) / max(1, int(prof[8 + i])):.0f} cyc x {int(prof[8 + i]) // 128})"
                                for i, nm in enumerate(names)), f"; {tot / 128 / len(data):.1f} cyc per output byte")
    print(f"{label}: {len(data)} B from {plen} B of payload, block lifetime {us:.0f} us = {us * 1e3 / len(data):.1f} ns per output byte, "
          f"{us * 1e3 / (plen * 8):.2f} ns per payload bit")
