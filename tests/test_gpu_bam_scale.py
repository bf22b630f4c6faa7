"""
GPU: BASELINE config 5 at real size -- a > 4 GiB, three-contig 60x coordinate-sorted paired-end BAM (a chr1-sized
contig between two small ones, 100.9 M records) streamed through the device inflate + device record parser
(``source.stream_source``), every feature of every 100 kb window and per-base WPS, against the C oracle in
read1-fetch mode (reference ``io/alignment.py:242-268``: a BAM window query returns read1 alignments overlapping the
window).  What this guards that the 24 Mb slice of tests/test_gpu_config45.py cannot: 64-bit file offsets, ring-slot
reuse over thousands of pieces, the hand-over across contig runs, and BAI region reads BEHIND the 4 GiB mark.

``FTK_BIG_BAM_BP`` (default: chr1 of b37, 249 250 621) sizes the middle contig; anything that keeps the file above
4 GiB runs the full set of assertions, a smaller value (developer boxes) skips the 4 GiB ones.
"""
import os

import numpy as np
import pytest

from finaletoolkit_amd import synth
from oracle import scale_check as SC

pytestmark = pytest.mark.gpu

BIG = int(os.environ.get("FTK_BIG_BAM_BP", synth.B37_SIZES["1"]))
CONTIGS = [("small_a", 3_000_000), ("big", BIG), ("small_c", 5_000_000)]
GIB4 = 1 << 32


@pytest.fixture(scope="module")
def big_bam(tmp_path_factory):
    d = tmp_path_factory.mktemp("bigbam")
    path = str(d / "wg60x.bam")
    exp = synth.write_paired_bam_native(path, CONTIGS, 60.0, 4242)  # (same records as write_paired_bam_contigs, written in C)
    yield path, exp
    from finaletoolkit_amd import source
    source.close_all()
    for p in (path, path + ".bai"):
        if os.path.exists(p):
            os.remove(p)


def test_config5_bam_at_real_size_streams_through_the_device_parser(big_bam):
    from finaletoolkit_amd import source
    path, exp = big_bam
    full = BIG >= 200_000_000
    file_bytes = os.path.getsize(path)
    if full:
        assert file_bytes > GIB4 and exp["small_c"]["first_off"] > GIB4
        assert sum(v["n"] for v in exp.values()) == 600_000 + 49_850_124 + 1_000_000
    source.close_all()
    eng = source.get_engine()
    seen, whole = [], {}
    for src, name in source.stream_source(path):
        size = dict(CONTIGS)[name]
        key = src.key(name)
        assert eng.is_bam(key)
        ws, we = synth.tiling_windows(size, SC.WINDOW)
        f = eng.window_features(key, ws, we, 30, hist=(0, 1001), delfi=dict(quality_threshold=30))
        ok, detail = SC.check_contig(eng, key, size, exp[name], f, n_sampled=24)
        assert ok, (name, detail)
        assert detail["windows_checked"] >= 24 and detail["wps_bases_checked"] >= 150_000
        # every base of the contig in one launch, against the closed form of its sum (every read1 overlaps the
        # contig-wide fetch window, so read1 semantics drop nothing here)
        w = eng.wps(key, 0, size, size, 120, 120, 180, 30)
        assert len(w) == size and int(w.sum()) == SC.wps_closed_form_sum(exp[name], size), name
        whole[name] = (f, w)
        seen.append(name)
    assert seen == [c for c, _ in CONTIGS]
    stage = src.decode_stage_ms
    assert stage is not None and stage["inflate"] >= 0
    # ---- region reads through the BAI's linear index, both BEHIND the 4 GiB mark --------------------------------
    source.close_all()
    del source.REGION_READS[:]
    lazy = source.open_source(path)
    eng = source.get_engine()                      # (close_all dropped the engine the stream used)
    assert lazy.lazy and not lazy.loaded
    regions = [("small_c", 2_000_000, 2_400_000), ("big", BIG * 24 // 25 // SC.WINDOW * SC.WINDOW, BIG * 24 // 25 // SC.WINDOW * SC.WINDOW + 400_000),
               ("small_a", 1_000_000, 1_300_000)]
    for name, a, b in regions:
        size = dict(CONTIGS)[name]
        off = SC.region_file_offset(exp[name], a)
        if full and name != "small_a":
            assert off > GIB4, (name, off)
        key = lazy.require_region(name, a, b)
        assert key in lazy.regions and name not in lazy.loaded
        ok, detail = SC.check_region(eng, key, size, exp[name], a, b)
        assert ok, (name, a, b, detail)
        assert detail["region_rows"] < detail["contig_rows"] // 5  # a region, not the contig
        # ... and equal to the streamed whole-contig answer on the same windows / bases
        f, w = whole[name]
        ws = np.arange(a, b, SC.WINDOW, dtype=np.int32)
        g = eng.window_features(key, ws, (ws + SC.WINDOW).astype(np.int32), 30, hist=(0, 1001),
                                delfi=dict(quality_threshold=30))
        i0 = a // SC.WINDOW
        for k in ("coverage", "hist", "overflow", "short", "long"):
            assert np.array_equal(g[k], f[k][i0:i0 + len(ws)]), (name, k)
        lazy.release_region(key)
    assert [r[1] for r in source.REGION_READS] == [r[0] for r in regions]
    source.close_all()
