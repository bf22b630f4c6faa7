"""GPU, BASELINE configs 4 and 5 at their own sizes.

Config 4: whole-genome synthetic 30x (b37 1-22,X,Y: 309.6 M fragments generated on the device), DELFI short /
long counts of all 30 970 x 100 kb bins with blacklist and gap intervals, merged to 5 Mb arm windows
(frag/_delfi.py:269-370, frag/_delfi_merge_bins.py).  Bit-exact against the C oracle on three whole contigs
(chr1 among them: 24.9 M fragments, 2 493 windows in one launch) and through size-independent identities on
the whole genome.

Config 5: a 60x coordinate-sorted paired-end BAM of one contig through the streaming decoder
(decode || H2D || kernels), then all features in one sweep -- coverage + 1001-bin length histogram + DELFI
counts + per-base WPS, fused and as separate calls -- bit-exact against the oracle in read1-fetch mode
(io/alignment.py:242-268)."""
import numpy as np
import pandas
import pytest

from finaletoolkit_amd import synth
from oracle import oracle as O

pytestmark = pytest.mark.gpu
WINDOW = 100_000


# ---- config 4 -------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def genome(engine):
    """Every b37 contig at 30x resident in HBM (3.1 GB of columns), its bins, blacklist and gap constants, and
    the DELFI counts of all bins computed contig by contig."""
    import torch
    dev = torch.device("cuda", 0)
    per = {}
    for ci, (c, size) in enumerate(synth.B37_SIZES.items()):
        n = synth.n_fragments(size, 30.0)
        s, e, q, st = synth.gen_contig_device(torch, dev, size, n, synth.SEED_BASE + ci)
        torch.cuda.synchronize()
        engine.load_contig_device(f"g4:{c}", s, e, q, st, n)
        # per-contig facts for the identities, computed with torch on the generator's own tensors
        ln = (e - s).to(torch.int64)
        ok = (q >= 30) & (ln >= 100) & (ln <= 220)
        per[c] = dict(size=size, n=n, cols=(s, e, q, st) if c in ("1", "9", "Y") else None,
                      n_delfi=int(ok.sum().item()), n_long=int((ok & (ln >= 151)).sum().item()))
        ws, we = synth.tiling_windows(size, WINDOW)
        per[c]["ws"], per[c]["we"] = ws, we
        per[c]["bl"] = synth.synth_blacklist(size, 77 + ci, max(8, int(2000 * size / 3.1e9)))
        per[c]["gaps"] = synth.synth_gaps(size)
        del s, e, q, st, ln, ok
    yield per
    for c in synth.B37_SIZES:
        engine.release(f"g4:{c}")


def test_config4_whole_genome_delfi_bins(engine, genome):
    total_bins = sum(len(p["ws"]) for p in genome.values())
    assert total_bins == 30_970 and sum(p["n"] for p in genome.values()) == 309_567_743
    rows = []
    plain_short = plain_long = 0
    for c, p in genome.items():
        sh, lg, nf = engine.delfi_counts(f"g4:{c}", p["ws"], p["we"], 30, p["bl"][0], p["bl"][1], p["gaps"])
        assert np.array_equal(sh + lg, nf)
        p["counts"] = (sh, lg)
        # without blacklist and gaps every DELFI-sized mapq>=30 fragment lands in exactly one bin of the tiling
        sh0, lg0, _ = engine.delfi_counts(f"g4:{c}", p["ws"], p["we"], 30)
        assert int(lg0.sum()) == p["n_long"] and int(sh0.sum()) == p["n_delfi"] - p["n_long"], c
        assert np.all(sh <= sh0) and np.all(lg <= lg0)
        # bins inside the centromere interval lose everything; bins clear of both intervals and of every
        # blacklist region lose nothing
        c0, c1, tel = p["gaps"]
        inside = (p["ws"] >= c0) & (p["we"] <= c1)
        assert inside.sum() >= 29 and sh[inside].sum() == 0 and lg[inside].sum() == 0
        far = (p["we"].astype(np.int64) + 1000 < c0) | (p["ws"].astype(np.int64) - 1000 > c1)
        far &= (p["ws"] > tel[0][1] + 1000) & (p["we"] < tel[1][0] - 1000)
        bs, be = p["bl"]
        hit = np.zeros(len(p["ws"]), bool)
        hit[np.clip(bs // WINDOW, 0, len(hit) - 1)] = True
        hit[np.clip((be - 1) // WINDOW, 0, len(hit) - 1)] = True
        clean = far & ~hit
        assert clean.sum() > len(clean) // 2
        assert np.array_equal(sh[clean], sh0[clean]) and np.array_equal(lg[clean], lg0[clean]), c
        plain_short += int(sh0.sum())
        plain_long += int(lg0.sum())
        # the frame frag.delfi builds per bin (arm labels as its gap annotation would give: p left of the
        # centromere interval, q right of it, NOARM inside)
        arm = np.where(p["we"] <= c0, c + "p", np.where(p["ws"] >= c1, c + "q", "NOARM"))
        for i in range(len(sh)):
            if arm[i] != "NOARM":
                rows.append((c, int(p["ws"][i]), int(p["we"][i]), arm[i], int(sh[i]), int(lg[i]), 0.41, int(nf[i]),
                             sh[i] / lg[i] if lg[i] else np.nan))
    assert plain_short + plain_long == sum(p["n_delfi"] for p in genome.values())
    # 100 kb -> 5 Mb (frag/_delfi_merge_bins.py): every merged window is the sum of fifty consecutive bins of
    # one arm, p-arms anchored at their first bin, q-arms at their last
    from finaletoolkit_amd.frag import delfi_merge_bins
    df = pandas.DataFrame(rows, columns=["contig", "start", "stop", "arm", "short", "long", "gc", "num_frags", "ratio"])
    merged = delfi_merge_bins(df, gc_corrected=False)
    assert 540 <= len(merged) <= 633  # 633 whole 5 Mb bins tile the genome; arms and centromeres cost a few per contig
    by_arm = {a: g for a, g in df.groupby("arm", sort=False)}
    for arm, g in merged.groupby("arm", sort=False):
        src = by_arm[arm]
        skip = len(src) % 50 if "q" in arm else 0
        assert len(g) == len(src) // 50
        for k, (_, r) in enumerate(g.iterrows()):
            part = src.iloc[skip + 50 * k: skip + 50 * (k + 1)]
            assert (r["start"], r["stop"]) == (part["start"].min(), part["stop"].max())
            assert (r["short"], r["long"], r["num_frags"]) == (part["short"].sum(), part["long"].sum(), part["num_frags"].sum())
    # and the fused whole-genome launch (ftk_window_features_batch: one launch for all 30 970 bins, what a rank
    # of the sharded run issues) gives the same vector
    import torch
    dev = torch.device("cuda", 0)
    items = [dict(name=f"g4:{c}", starts=p["ws"], stops=p["we"], bl_start=p["bl"][0], bl_end=p["bl"][1], gaps=p["gaps"])
             for c, p in genome.items()]
    batch = engine.feature_batch(items, 30)
    d_sh = torch.zeros(total_bins, dtype=torch.int64, device=dev)
    d_lg = torch.zeros(total_bins, dtype=torch.int64, device=dev)
    engine.window_features_batch(batch, delfi_q=30, short=d_sh, long=d_lg)
    engine.sync()
    assert np.array_equal(d_sh.cpu().numpy(), np.concatenate([p["counts"][0] for p in genome.values()]))
    assert np.array_equal(d_lg.cpu().numpy(), np.concatenate([p["counts"][1] for p in genome.values()]))


@pytest.mark.parametrize("contig", ["1", "9", "Y"])
def test_config4_whole_contigs_against_the_oracle(engine, genome, contig):
    p = genome[contig]
    s, e, q, st = (t.cpu().numpy() for t in p["cols"])
    fr = O.Frags(s, e, q, st)
    want = O.c_delfi_counts(fr, p["ws"], p["we"], 30, p["bl"][0], p["bl"][1], p["gaps"])
    got = engine.delfi_counts(f"g4:{contig}", p["ws"], p["we"], 30, p["bl"][0], p["bl"][1], p["gaps"])
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    if contig == "1":
        assert len(s) == 24_925_062 and len(p["ws"]) == 2_493
        # the same contig through the fused pass (coverage + 1001-bin histogram + DELFI in one sweep)
        f = engine.window_features(f"g4:{contig}", p["ws"], p["we"], 30, hist=(0, 1001),
                                   delfi=dict(quality_threshold=30, bl_start=p["bl"][0], bl_end=p["bl"][1], gaps=p["gaps"]))
        assert np.array_equal(f["short"], want[0]) and np.array_equal(f["long"], want[1])
        assert np.array_equal(f["coverage"], O.c_window_counts(fr, p["ws"], p["we"], mapq_min=30))
        h, o = O.c_fraglen_hist(fr, p["ws"], p["we"], 0, 1001, mapq_min=30)
        assert np.array_equal(f["hist"], h) and np.array_equal(f["overflow"], o)
        # and the whole contig's WPS (249 M bases in one launch) on sampled tiles
        import torch
        out = torch.empty(p["size"], dtype=torch.int64, device="cuda:0")
        engine.wps(f"g4:{contig}", 0, p["size"], p["size"], 120, 120, 180, 30, out=out)
        engine.sync()
        rng = np.random.default_rng(4)
        for a in [0, p["size"] - 5000] + [int(x) for x in rng.integers(0, p["size"] - 5000, 6)]:
            assert np.array_equal(out[a:a + 5000].cpu().numpy(), O.c_wps(fr, a, a + 5000, p["size"], 120, 120, 180, 30)), a
        # sum over all 249 M bases: a passing fragment gives 1 to [fs+61, fe-60] and takes 1 from [fs-59, fs+60] and
        # [fe-59, fe+60], each clipped to the contig
        keep = (q >= 30) & (e - s >= 120) & (e - s <= 180)
        fs, fe = s[keep].astype(np.int64), e[keep].astype(np.int64)
        size = p["size"]

        def clipped(a, b):  # bases of the closed range [a, b] inside [0, size)
            return np.maximum(np.minimum(b, size - 1) - np.maximum(a, 0) + 1, 0)
        want_sum = int((clipped(fs + 61, fe - 60) - clipped(fs - 59, fs + 60) - clipped(fe - 59, fe + 60)).sum())
        assert int(out.sum().item()) == want_sum


# ---- config 5 -------------------------------------------------------------------------------------------
def test_config5_bam_60x_stream_all_features(engine, tmp_path):
    from finaletoolkit_amd import source
    size, contig = 24_000_000, "mid"
    bam = tmp_path / "mid60x.bam"
    exp = synth.write_paired_bam(str(bam), contig, size, 60.0, 31)
    assert exp["n"] == 4_800_000
    got_contigs = []
    for src, name in source.stream_source(str(bam)):
        got_contigs.append(name)
        key = src.key(name)
        assert engine is not None and source.get_engine().is_bam(key)
    assert got_contigs == [contig]
    eng = source.get_engine()
    n, max_len, _ = eng.info(key)
    assert n == exp["n"]
    fr = O.Frags(exp["s"], exp["e"], exp["q"], exp["st"], exp["r1s"], exp["r1e"])
    ws, we = synth.tiling_windows(size, WINDOW)
    bl = synth.synth_blacklist(size, 5, 60)
    gaps = synth.synth_gaps(size)
    # one sweep: coverage + histogram + DELFI (read1-overlap fetch semantics: a fragment counts in a window only
    # if its read1 alignment overlaps it -- fragments whose read1 lies in the neighbouring bin drop out)
    f = eng.window_features(key, ws, we, 30, hist=(0, 1001),
                            delfi=dict(quality_threshold=30, bl_start=bl[0], bl_end=bl[1], gaps=gaps))
    want_c = O.c_window_counts(fr, ws, we, mapq_min=30)
    h, o = O.c_fraglen_hist(fr, ws, we, 0, 1001, mapq_min=30)
    want_d = O.c_delfi_counts(fr, ws, we, 30, bl[0], bl[1], gaps)
    assert np.array_equal(f["coverage"], want_c) and np.array_equal(f["hist"], h) and np.array_equal(f["overflow"], o)
    assert np.array_equal(f["short"], want_d[0]) and np.array_equal(f["long"], want_d[1])
    tab = O.Frags(exp["s"], exp["e"], exp["q"], exp["st"])  # the same fragments under tabix semantics
    assert int(want_c.sum()) < int(O.c_window_counts(tab, ws, we, mapq_min=30).sum())  # read1 semantics do bite
    # per-base WPS: engine and oracle on the SAME intervals (a BAM fetch also asks read1 to overlap the interval's
    # fetch window, so a tile is not always a slice of the whole-contig call) ...
    rng = np.random.default_rng(8)
    for a in [0, size - 5000] + [int(x) for x in rng.integers(0, size - 5000, 12)]:
        assert np.array_equal(eng.wps(key, a, a + 5000, size, 120, 120, 180, 30),
                              O.c_wps(fr, a, a + 5000, size, 120, 120, 180, 30)), a
    a = 7_000_000
    assert np.array_equal(eng.wps(key, a, a + 200_000, size), O.c_wps(fr, a, a + 200_000, size))  # (the oracle is quadratic in the interval: 1.5 Mb took 200 s of the suite)
    # ... and the whole contig (24 M bases in one launch; every read1 overlaps the contig-wide fetch window)
    # against the closed form: a passing fragment takes 1 from [fs-59, fs+60] and [fe-59, fe+60] and gives 1 to
    # the bases between
    whole = eng.wps(key, 0, size, size, 120, 120, 180, 30)
    keep = (exp["q"] >= 30) & (exp["e"] - exp["s"] >= 120) & (exp["e"] - exp["s"] <= 180)
    fs, fe = exp["s"][keep].astype(np.int64), exp["e"][keep].astype(np.int64)
    d = np.zeros(size + 400, np.int64)
    for pos, val in ((fs - 59, -1), (fs + 61, 2), (fe - 59, -2), (fe + 61, 1)):
        np.add.at(d, pos + 100, val)
    assert np.array_equal(whole, np.cumsum(d)[100:100 + size])
    source.close_all()
