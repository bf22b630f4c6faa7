"""CPU: bigWig writer/reader (finaletoolkit_amd/bigwig.py).  The reader is checked on a
file pyBigWig wrote (the reference's tests/data/test.bw); the writer by round trip."""
import os

import numpy as np

from finaletoolkit_amd.bigwig import read_bigwig, write_fixed_step_bigwig
from tests.helpers import DATA


def test_reader_on_pybigwig_file():
    chroms, iv = read_bigwig(os.path.join(DATA, "test.bw"))
    assert chroms == {"chr1": (0, 1000000)}
    assert iv == [("chr1", 1000 + i, 1001 + i, float(i)) for i in range(5)]


def test_roundtrip_sections_and_order(tmp_path, capsys):
    hdr = [("2", 100_000_000), ("10", 100_000_000), ("chrUn_long_name", 5000)]
    vals = [("2", 1000, np.arange(-50, 40_000)), ("10", 5, np.array([3, -7, 0])),
            ("2", 10, np.array([1, 2])),           # out of order: skipped like pyBigWig's RuntimeError path
            ("chrUn_long_name", 10, np.arange(5)), ("nope", 0, np.array([1]))]
    p = str(tmp_path / "t.bw")
    write_fixed_step_bigwig(p, hdr, iter(vals))
    chroms, iv = read_bigwig(p)
    assert chroms == {"2": (0, 100_000_000), "10": (1, 100_000_000), "chrUn_long_name": (2, 5000)}
    kept = [vals[0], vals[1], vals[3]]
    assert iv == [(c, s + i, s + i + 1, float(v)) for c, s, a in kept for i, v in enumerate(a)]
    assert "out of order" in capsys.readouterr().err


def test_roundtrip_many_contigs_multilevel_trees(tmp_path):
    hdr = [(f"c{i}", 10 ** 6) for i in range(700)]
    vals = [(f"c{i}", 100 * i, np.full(3, i)) for i in range(700)]
    p = str(tmp_path / "t2.bw")
    write_fixed_step_bigwig(p, hdr, iter(vals))
    chroms, iv = read_bigwig(p)
    assert len(chroms) == 700 and chroms["c699"] == (699, 10 ** 6)
    assert iv == [(c, s + i, s + i + 1, float(v)) for c, s, a in vals for i, v in enumerate(a)]
