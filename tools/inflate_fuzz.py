"""An extended campaign of tests/test_gpu_inflate.py's random mixtures against the device inflate (all three symbol
loops against each other, then against the input): SEEDS x 300 BGZF blocks of random make-up through random zlib levels
and strategies, plus the make-ups that press on the lane-parallel loop's limits - one dominant symbol (1-bit codes:
hundreds of tokens per stretch, more than a lane may write), near-uniform code lengths (chains that never fall into
step), maximal matches at distance 1, and blocks whose data ends within bytes of a stretch's end.
usage: python3 tools/inflate_fuzz.py [seeds=20]"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tests.test_gpu_inflate as T  # noqa: E402
from finaletoolkit_amd import bgzf  # noqa: E402
from finaletoolkit_amd.engine import Engine  # noqa: E402

n_seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
eng = Engine(0)
STRATEGIES = [zlib.Z_DEFAULT_STRATEGY, zlib.Z_FILTERED, zlib.Z_RLE, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY]
DIST = [1, 2, 3, 7, 64, 258, 700, 1_700, 1_726, 1_727, 2_048, 2_049, 3_775, 4_096, 4_097, 9_000, 32_768, 40_000]
bad = 0
total = 0
for seed in range(1000, 1000 + n_seeds):
    rng = np.random.default_rng(seed)
    members, want = [], []
    while len(members) < 300:
        out = bytearray()
        target = int(rng.integers(1, 0xFF00))
        while len(out) < target:
            kind = int(rng.integers(0, 9))
            n = int(rng.integers(1, 6000))
            if kind == 0:
                out += rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            elif kind == 1:
                out += rng.integers(48, 58, n, dtype=np.uint8).tobytes()
            elif kind == 2:
                out += bytes([int(rng.integers(0, 256))]) * n
            elif kind == 3:  # one dominant symbol, a rare second and third: 1- and 2-bit codes
                p = float(rng.choice([0.9, 0.97, 0.995]))
                out += rng.choice(np.array([65, 66, 67], np.uint8), n, p=[p, (1 - p) * 0.7, (1 - p) * 0.3]).tobytes()
            elif kind == 4:  # 128 or 64 equally likely symbols: every code the same length
                k = int(rng.choice([64, 128]))
                out += rng.integers(0, k, n, dtype=np.uint8).tobytes()
            elif kind == 5 and out:  # a maximal run: matches of 258 at distance 1
                out += bytes([out[-1]]) * int(rng.integers(258, 5000))
            elif kind == 6:  # text-like rows
                rows = [b"chr%d\t%d\t%d\t%d\t%s\n" % (seed % 22 + 1, int(a), int(a) + int(rng.integers(30, 400)), int(rng.integers(0, 61)), (b"+", b"-")[int(rng.integers(0, 2))])
                        for a in np.sort(rng.integers(0, 10**8, max(1, n // 30)))]
                out += b"".join(rows)
            elif out:
                d = int(min(len(out), rng.choice(DIST)))
                for _ in range(n // d + 1):
                    out += out[len(out) - d:len(out) - d + min(d, n)]
        data = bytes(out[:target])
        m = T._member(data, int(rng.integers(0, 10)), STRATEGIES[int(rng.integers(0, 5))])
        if m is not None:
            members.append(m)
            want.append(data)
    rc, got = T._inflate(eng, b"".join(members) + bgzf._EOF)
    expect = b"".join(want)
    total += len(members)
    if rc != 0 or got != expect:
        bad += 1
        off = 0
        first = next((k for k, w in enumerate(want) if got[sum(map(len, want[:k])):sum(map(len, want[:k + 1]))] != w), None) if rc == 0 else None
        print(f"seed {seed}: rc {rc}, first differing block {first}: {eng.lib.ftk_last_error(eng.ctx)!r}", flush=True)
print(f"{total} blocks of {n_seeds} seeds through the three symbol loops: {bad} seeds failed")
sys.exit(1 if bad else 0)
