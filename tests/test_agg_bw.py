"""CPU: ``agg_bw`` (utils/_agg_bw.py:18-146) - the reference's own known answers on its test.bw fixture and
the golden cases recorded from the imported reference (oracle/gen_golden_aggbw.py): returned arrays, dtypes
and the WIG text, through the product's bigWig reader."""
import io
import json
import os
from contextlib import redirect_stdout

import numpy as np
import pytest

from finaletoolkit_amd.utils import agg_bw
from tests.helpers import DATA, GOLDEN


def test_reference_known_answers(tmp_path):
    # tests/test_agg_bw.py:10-27 of the reference
    bw, bed = os.path.join(DATA, "test.bw"), os.path.join(DATA, "bw_test.bed")
    with redirect_stdout(io.StringIO()):
        assert agg_bw(bw, bed, tmp_path / "out.wig", 0) == pytest.approx([0., 0., 0., 0., 0.])
    assert agg_bw(bw, bed, tmp_path / "out.wig", 2) == pytest.approx([1., 2., 3.])
    with pytest.raises(ValueError):
        agg_bw(bw, os.path.join(DATA, "b37.chrom.sizes"), tmp_path / "out.wig")
    with pytest.raises(ValueError):
        agg_bw(bw, bed, tmp_path / "out.txt", 2)


def test_golden_cases_from_the_reference(tmp_path):
    gold = np.load(os.path.join(GOLDEN, "aggbw.npz"))
    meta = json.load(open(os.path.join(GOLDEN, "aggbw.json")))
    assert len(meta) == 5
    for cs in meta:
        out = str(tmp_path / (cs["key"] + ".wig"))
        with redirect_stdout(io.StringIO()) as printed, np.errstate(all="ignore"):
            got = agg_bw(os.path.join(GOLDEN, "aggbw_track.bw"), os.path.join(GOLDEN, "aggbw_sites.bed"), out, **cs["kwargs"])
        want = gold[cs["key"]]
        assert str(got.dtype) == cs["dtype"] and got.shape == want.shape, cs["key"]
        assert np.array_equal(got, want, equal_nan=True), cs["key"]      # sums of float32-valued entries: exact
        assert open(out).read() == cs["wig"], cs["key"]
        assert len(printed.getvalue().splitlines()) == cs["printed_lines"], cs["key"]  # the same intervals were skipped


def test_cli_agg_bw(tmp_path):
    from finaletoolkit_amd import cli
    out = str(tmp_path / "agg.wig")
    with redirect_stdout(io.StringIO()):
        cli.main(["agg-bw", os.path.join(GOLDEN, "aggbw_track.bw"), os.path.join(GOLDEN, "aggbw_sites.bed"), "-o", out,
                  "-m", "120"])
    meta = {m["key"]: m for m in json.load(open(os.path.join(GOLDEN, "aggbw.json")))}
    assert open(out).read() == meta["w120"]["wig"]
