"""
GPU: a fragment file LARGER THAN 4 GiB -- the whole b37 genome (1-22, X, Y) at 80x as one tabix-indexed frag.gz
(825 M rows, 21 GB of text, ~5.1 GB compressed) streamed through the device inflate + device row parser
(``source.stream_source``; reference ``io/alignment.py:270-302`` over ``utils/_frag_generator.py:58-141``), and region
reads through the ``.tbi`` linear index whose file offsets lie BEHIND the 4 GiB mark (contigs X and Y start there).
The fragment-file twin of tests/test_gpu_bam_scale.py: what the 30x genome of ``bench.py`` (2 GB) cannot show is that
nothing on the text path keeps a file offset, a virtual offset or a piece count in 32 bits.

``FTK_BIG_TEXT_DEPTH`` (default 80) sizes the file; a smaller value (developer boxes) runs the same assertions
without the 4 GiB ones.
"""
import os

import numpy as np
import pytest

from finaletoolkit_amd import bgzf, synth
from oracle import scale_check as SC

pytestmark = pytest.mark.gpu

DEPTH = float(os.environ.get("FTK_BIG_TEXT_DEPTH", "80"))
NAMES = list(synth.B37_SIZES)
KEPT = ("1", "X", "Y")          # contigs whose fragments stay in memory for the oracle (the others: exact counts only)
GIB4 = 1 << 32


@pytest.fixture(scope="module")
def big_text(tmp_path_factory):
    import torch
    d = tmp_path_factory.mktemp("bigtext")
    path = str(d / "wg80x.frag.gz")
    dev = torch.device("cuda:0")
    exp = {}

    def contigs():
        for k, c in enumerate(NAMES):
            size = synth.B37_SIZES[c]
            n = synth.n_fragments(size, DEPTH)
            s, e, q, st = (t.cpu().numpy() for t in synth.gen_contig_device(torch, dev, size, n, 991 + k))
            exp[c] = dict(n=n, cov=int((q >= 30).sum()))
            if c in KEPT:
                exp[c].update(s=s, e=e, q=q, st=st)
            yield c, s, e, q, st

    where = bgzf.write_frag_gz_contigs(path, contigs(), level=1, with_index=True)
    torch.cuda.empty_cache()
    for c in NAMES:
        exp[c].update(where[c])
    yield path, exp
    from finaletoolkit_amd import source
    source.close_all()
    for p in (path, path + ".tbi"):
        if os.path.exists(p):
            os.remove(p)


def test_frag_file_beyond_4gib_streams_and_serves_regions(big_text):
    from finaletoolkit_amd import source
    path, exp = big_text
    full = DEPTH >= 78
    if full:
        assert os.path.getsize(path) > GIB4 and exp["X"]["first_off"] > GIB4 and exp["Y"]["first_off"] > GIB4
    source.close_all()
    eng = source.get_engine()
    seen, whole = [], {}
    for src, name in source.stream_source(path):
        size = synth.B37_SIZES[name]
        key = src.key(name)
        assert not eng.is_bam(key)
        assert int(eng.info(key)[0]) == exp[name]["n"], name
        ws, we = synth.tiling_windows(size, SC.WINDOW)
        f = eng.window_features(key, ws, we, 30, hist=(0, 1001), delfi=dict(quality_threshold=30))
        # every fragment's midpoint lies in exactly one tiling window: the contig's coverage sums to its mapq >= 30 rows
        assert int(f["coverage"].sum()) == exp[name]["cov"] == int(f["hist"].sum() + f["overflow"].sum()), name
        if name in KEPT:
            ok, detail = SC.check_contig(eng, key, size, exp[name], f, n_sampled=24)
            assert ok, (name, detail)
            w = eng.wps(key, 0, size, size, 120, 120, 180, 30)
            assert len(w) == size and int(w.sum()) == SC.wps_closed_form_sum(exp[name], size), name
            whole[name] = f
            del w
        seen.append(name)
    assert seen == NAMES
    # ---- region reads through the tabix linear index; X and Y lie BEHIND the 4 GiB mark ------------------------------
    source.close_all()
    del source.REGION_READS[:]
    lazy = source.open_source(path)
    eng = source.get_engine()
    assert lazy.lazy and not lazy.loaded
    regions = [("Y", 20_000_000, 20_400_000), ("X", 150_000_000, 150_500_000), ("1", 100_000_000, 100_300_000)]
    for name, a, b in regions:
        size = synth.B37_SIZES[name]
        off = int(exp[name]["linear"][a >> 14] >> np.uint64(16))
        assert exp[name]["first_off"] <= off < exp[name]["end_off"]
        if full and name != "1":
            assert off > GIB4, (name, off)
        key = lazy.require_region(name, a, b)
        assert key in lazy.regions and name not in lazy.loaded
        ok, detail = SC.check_region(eng, key, size, exp[name], a, b)
        assert ok, (name, a, b, detail)
        assert detail["region_rows"] < detail["contig_rows"] // 20  # a region, not the contig
        ws = np.arange(a, b, SC.WINDOW, dtype=np.int32)
        g = eng.window_features(key, ws, (ws + SC.WINDOW).astype(np.int32), 30, hist=(0, 1001),
                                delfi=dict(quality_threshold=30))
        i0 = a // SC.WINDOW
        for k in ("coverage", "hist", "overflow", "short", "long"):
            assert np.array_equal(g[k], whole[name][k][i0:i0 + len(ws)]), (name, k)
        lazy.release_region(key)
    assert [r[1] for r in source.REGION_READS] == [r[0] for r in regions]
    source.close_all()
