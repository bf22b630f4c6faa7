"""CPU: the native output writers (csrc/ftk_writers.cpp) produce the bytes the reference's Python writers
produce (frag/_wps.py:208-229, frag/_multi_wps.py:300-341): str(int) / repr(float) per base, gzip output that
decompresses to the same stream, bigWig data sections that the reader decodes to the same float32 values."""
import gzip
import struct
import zlib

import numpy as np
import pytest

from finaletoolkit_amd import bigwig, writers


def _ints(rng, n):
    v = rng.integers(-300, 300, n).astype(np.int64)
    v[:8] = [0, -1, 1, 9, 10, -10, 2 ** 63 - 1, -2 ** 63]
    return v


def test_wig_body_equals_python_join():
    rng = np.random.default_rng(1)
    for n in (0, 1, 7, 70_000, 300_001):
        v = _ints(rng, max(n, 8))[:n]
        with writers.wig_body(v) as b:
            assert b.tobytes() == "".join(f"{x}\n" for x in v).encode()
    with writers.wig_body(v, threads=3) as b:
        assert b.tobytes() == "".join(f"{x}\n" for x in v).encode()


def test_bedgraph_rows_int_runs():
    rng = np.random.default_rng(2)
    lens = [5000, 1, 0, 4999, 70_000, 3]
    starts = [0, 9_000, 20_000, 30_000, 2 ** 31 + 5, 2 ** 31 + 80_000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    v = _ints(rng, int(offs[-1]))
    want = "".join(f"chr1_x\t{s + i}\t{s + i + 1}\t{v[offs[k] + i]}\n" for k, s in enumerate(starts) for i in range(lens[k]))
    for threads in (0, 1, 5):
        with writers.bedgraph_rows("chr1_x", starts, v, offs, threads=threads) as rows:
            assert rows.tobytes() == want.encode()
    with writers.bedgraph_rows("c", 17, v[:10]) as rows:  # one run, scalar start
        assert rows.tobytes() == "".join(f"c\t{17 + i}\t{18 + i}\t{v[i]}\n" for i in range(10)).encode()


def test_float_rows_print_like_python():
    rng = np.random.default_rng(3)
    special = [0.0, -0.0, 1.0, -1.0, 100.0, 1e15, 1e16, 9.999999999999999e15, 1e17, 1e-4, 1e-5, 9.999e-5, 0.1, 1 / 3,
               33.333333333333336, 2.5e-10, 1.7976931348623157e308, 5e-324, 2.2250738585072014e-308, 123456789012345678.0,
               float("inf"), float("-inf"), float("nan"), 66.66666666666667, 0.30000000000000004, 1e22, 1e23, 123456.789]
    bits = rng.integers(0, 2 ** 63, 20_000, dtype=np.int64).view(np.float64)  # every exponent
    pct = rng.integers(0, 1000, 20_000) / rng.integers(1, 1000, 20_000) * 100  # cleavage-like percentages
    v = np.concatenate([special, bits, -bits[:100], pct])
    with writers.bedgraph_rows("c", 5, v) as rows:
        got = rows.tobytes().decode().splitlines()
    assert len(got) == len(v)
    for i, (line, x) in enumerate(zip(got, v)):
        assert line == f"c\t{5 + i}\t{6 + i}\t{float(x)!r}", (i, x)
        assert line.split("\t")[3] == f"{x}", (i, x)  # numpy's float64 prints the same way (what the f-string sees)


def test_gzip_members_decompress_to_the_stream(tmp_path):
    rng = np.random.default_rng(4)
    v = _ints(rng, 900_000)
    p = tmp_path / "x.wig.gz"
    header = b"fixedStep\tchrom=1\tstart=0\tstep=1\tspan=900000\n"
    writers.write_text(p, header, writers.GZIP_LEVEL)
    with writers.wig_body(v) as body:
        body.write(p, writers.GZIP_LEVEL, append=True)
        raw = body.tobytes()
    assert len(raw) > 3 << 20  # several 1 MB members
    assert gzip.open(p, "rb").read() == header + raw
    q = tmp_path / "plain.wig"
    writers.write_text(q, header)
    with writers.wig_body(v) as body:
        body.write(q, 0, append=True)
    assert open(q, "rb").read() == header + raw
    e = tmp_path / "empty.gz"
    writers.write_text(e, b"", 6)
    assert gzip.open(e, "rb").read() == b""
    with pytest.raises(OSError):
        writers.write_text(tmp_path / "no" / "such" / "dir.gz", b"x", 6)


def test_bigwig_sections_match_the_python_statement():
    rng = np.random.default_rng(5)
    lens = [40_000, 5, 16_384, 16_385, 0, 1]
    starts = [10, 50_000, 60_000, 100_000, 200_000, 300_000]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    for values in (_ints(rng, int(offs[-1])) % 1000 - 500, rng.normal(0, 50, int(offs[-1]))):
        blob, table, stats = writers.bigwig_sections(3, starts, values, offs, 16384)
        want_tab, pos = [], 0
        for k, s in enumerate(starts):
            for o in range(0, lens[k], 16384):
                want_tab.append((s + o, s + min(o + 16384, lens[k])))
        assert [tuple(r[:2]) for r in table.tolist()] == want_tab
        for (s0, e0, size), st in zip(table.tolist(), stats):
            raw = zlib.decompress(blob[pos:pos + size])
            pos += size
            cid, a, b, step, span, typ, _, n = struct.unpack_from("<IIIIIBBH", raw, 0)
            assert (cid, a, b, step, span, typ, n) == (3, s0, e0, 1, 1, 3, e0 - s0)
            k = max(j for j, s in enumerate(starts) if s <= s0 and lens[j])
            src = np.asarray(values[offs[k] + (s0 - starts[k]):offs[k] + (e0 - starts[k])]).astype(np.float64).astype("<f4")
            assert raw[24:] == src.tobytes()
            c64 = src.astype(np.float64)
            assert st[0] == c64.min() and st[1] == c64.max()
            assert st[2] == pytest.approx(c64.sum(), rel=1e-12) and st[3] == pytest.approx((c64 * c64).sum(), rel=1e-12)
        assert pos == len(blob)


def test_bigwig_runs_writer_round_trip(tmp_path, capsys):
    rng = np.random.default_rng(6)
    header = [("chrA", 500_000), ("chrB", 300_000)]
    a_starts, a_lens = [1000, 20_000, 15_000, 90_000], [5000, 40_000, 100, 7]   # third one is out of order
    a_offs = np.concatenate([[0], np.cumsum(a_lens)]).astype(np.int64)
    a_vals = _ints(rng, int(a_offs[-1])) % 1000
    b_vals = rng.normal(0, 3, 12_345)
    path = tmp_path / "t.bw"
    bigwig.write_fixed_step_bigwig_runs(str(path), header, iter([("chrA", a_starts, a_vals, a_offs),
                                                                ("chrZ", [5], a_vals[:3], np.array([0, 3])),
                                                                ("chrB", [77], b_vals, np.array([0, len(b_vals)]))]))
    err = capsys.readouterr().err
    assert "chrA:15000-15100" in err and "chrZ:5-8" in err
    bw = bigwig.BigWigFile(str(path))
    assert bw.chroms() == dict(header)
    for k in (0, 1, 3):
        s, n = a_starts[k], a_lens[k]
        st, en, v = bw.intervals("chrA", s, s + n)
        assert np.array_equal(st, np.arange(s, s + n)) and np.array_equal(en, st + 1)
        assert np.array_equal(v, a_vals[a_offs[k]:a_offs[k + 1]].astype(np.float32).astype(np.float64))
    assert bw.intervals("chrA", 15_000, 15_100) is None
    st, en, v = bw.intervals("chrB", 0, 300_000)
    assert st[0] == 77 and len(v) == len(b_vals) and np.array_equal(v, b_vals.astype(np.float32).astype(np.float64))
    # the per-interval entry point is the same writer
    p2 = tmp_path / "t2.bw"
    bigwig.write_fixed_step_bigwig(str(p2), header, iter([("chrA", 1000, a_vals[:5000]), ("chrB", 77, b_vals)]))
    assert np.array_equal(bigwig.BigWigFile(str(p2)).values("chrB", 77, 77 + 100), b_vals[:100].astype(np.float32))


def test_bedgraph_batches_concatenate_to_the_whole():
    rng = np.random.default_rng(8)
    lens = [10, 0, 700, 5, 2500, 1, 1, 900]
    starts = [5, 100, 200, 2000, 3000, 9000, 9001, 9500]
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    v = _ints(rng, int(offs[-1]))
    with writers.bedgraph_rows("q", starts, v, offs) as rows:
        want = rows.tobytes()
    for budget in (1, 7, 600, 1000, 10 ** 9):
        got = b""
        for piece in writers.bedgraph_batches("q", starts, v, offs, max_values=budget):
            with piece:
                got += piece.tobytes()
        assert got == want, budget
