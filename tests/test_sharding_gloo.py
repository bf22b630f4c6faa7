"""
CPU, world_size 2 over gloo: the contig sharding + all-gather of bin vectors
reproduces the single-process result in contig order.  (Per-rank counts come
from the oracle here -- there is no GPU in this test -- the code under test is
finaletoolkit_amd.sharding.)
"""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from finaletoolkit_amd import sharding, synth
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sizes = {"a": 900_000, "b": 700_000, "c": 400_000, "d": 350_000, "e": 120_000}
    names = list(sizes)
    mine = sharding.shard_contigs(names, sizes, rank, world)
    n_bins, local, total = {}, {}, 0
    for i, c in enumerate(names):
        ws, we = synth.tiling_windows(sizes[c], 50_000)
        n_bins[c] = len(ws)
        if c in mine:
            s, e, qq, st = synth.synth_contig(sizes[c], depth=3.0, seed=50 + i)
            fr = O.Frags(s, e, qq, st)
            sh, lg, nf = O.c_delfi_counts(fr, ws, we, 30)
            local[c] = np.stack([sh, lg, nf], axis=1)
            total += int(O.c_window_counts(fr, [0], [None], mapq_min=30)[0])
    full = sharding.gather_bin_vectors(local, names, n_bins, sizes)
    grand = sharding.allreduce_sum(total)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, mine, {k: v.tolist() for k, v in full.items()}, grand))


def test_two_rank_gather_matches_single_process():
    import torch.multiprocessing as mp
    from finaletoolkit_amd import sharding, synth
    from oracle import oracle as O
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = {"a": 900_000, "b": 700_000, "c": 400_000, "d": 350_000, "e": 120_000}
    want, grand = {}, 0
    for i, c in enumerate(sizes):
        ws, we = synth.tiling_windows(sizes[c], 50_000)
        s, e, qq, st = synth.synth_contig(sizes[c], depth=3.0, seed=50 + i)
        fr = O.Frags(s, e, qq, st)
        want[c] = np.stack(O.c_delfi_counts(fr, ws, we, 30), axis=1).tolist()
        grand += int(O.c_window_counts(fr, [0], [None], mapq_min=30)[0])
    owned = []
    for rank, mine, full, g in res:
        assert full == want and g == grand
        owned += mine
    assert sorted(owned) == sorted(sizes)  # a partition of the contigs


def test_lpt_balance_on_b37():
    from finaletoolkit_amd import sharding, synth
    sizes = synth.B37_SIZES
    tot = sum(sizes.values())
    for world, ceiling in ((2, 0.99), (4, 0.98), (8, 0.95)):
        owner = sharding.lpt_assign(sizes, world)
        loads = [sum(sizes[c] for c in sizes if owner[c] == r) for r in range(world)]
        assert sum(loads) == tot and (tot / world) / max(loads) >= ceiling


# ---- the sharded product entry points (frag.delfi / frag.coverage), host logic on CPU -----------------------
# The device is replaced by a stand-in that answers the two engine calls from the oracle, so what runs here
# is exactly the rank logic of the product: contig dealing, per-rank work, all-gather / all-reduce, rank-0 writer.

def _fake_device(sizes):
    from finaletoolkit_amd import synth
    from oracle import oracle as O
    frs = {c: O.Frags(*synth.synth_contig(n, depth=3.0, seed=70 + i)) for i, (c, n) in enumerate(sizes.items())}
    asked = []
    regions = []

    class Src:
        contigs = list(sizes)
        loaded = set(sizes)
        region_reads = regions
        lengths = {c: None for c in sizes}

        def load_all(self):
            pass

        def has(self, c):
            return c in self.contigs

        def require(self, c):
            if c not in self.contigs:
                raise ValueError(f"could not create iterator for region '{c}'")
            asked.append(c)
            return c

        def key(self, c):
            return c

        def require_interval(self, c, *a, **k):
            return self.require(c)

        def require_region(self, c, a, b):  # (the whole contig's table answers every query of a region of it)
            regions.append((c, int(a), int(b)))
            return c

        def release_region(self, key):
            pass

    class Eng:
        def window_counts(self, name, starts, stops, q=30, lo=None, hi=None, policy="midpoint", out=None):
            return O.c_window_counts(frs[name], starts, stops, mapq_min=q, min_len=lo, max_len=hi, policy=policy)

        def delfi_counts(self, name, starts, stops, q=30, bs=None, be=None, gaps=None):
            return O.c_delfi_counts(frs[name], starts, stops, q, bs, be, gaps)

    class Ref:
        chroms = dict(sizes)

        def __init__(self, *_):
            pass

        def __enter__(self):
            return self

        def __exit__(self, *exc):
            pass

        def gc_counts(self, eng, contig, starts, stops):
            return (np.asarray(starts) // 7 + np.asarray(stops) % 11).astype(np.int64)

    return Src(), Eng(), Ref, asked


SIZES_P = {"a": 900_000, "b": 700_000, "c": 400_000, "d": 350_000, "e": 120_000}


def _product_worker(rank, world, port, d, q):
    sys.path.insert(0, ROOT)
    os.environ["FTK_SHARD_OVERHEAD_BASES"] = "0"  # (contigs under 1 Mb: the shares are cut by the intervals' span alone)
    import warnings
    from finaletoolkit_amd import sharding
    from finaletoolkit_amd.frag import _coverage as Cv, _delfi as Df
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          FTK_DIST_BACKEND="gloo")
        assert sharding.init_from_env() == (rank, world)
    src, eng, Ref, asked = _fake_device(SIZES_P)
    Cv.open_source = lambda *a, **k: src

    class Feed:  # source.ContigFeed over the stand-in: the one-process path of frag.coverage
        def __init__(self, *a, names=None, **k):
            self.names = list(src.contigs) if names is None else [c for c in names if src.has(c)]

        def __iter__(self):
            for c in self.names:
                asked.append(c)
                yield src, c

        def finish(self):
            return src

        def close(self):
            pass

    Cv.ContigFeed = Feed
    Df.resident_contigs = lambda path, names, *a, **k: ((src, c) for c in names if src.has(c))
    Df.region_contig = lambda path, c, lo, hi, *a, **k: (src, src.require(c))  # (the whole contig: a superset of the region)
    for mod in (Cv, Df):
        mod.get_engine = lambda: eng
    Df.ReferenceGenome = Ref
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        df = Df.delfi("x", f"{d}/cs.genome", f"{d}/bins.txt", "ref", blacklist_file=f"{d}/bl.bed", gap_file=f"{d}/gaps.bed",
                      no_gc_correct=True, remove_nocov=False, merge_bins=False, output_file=f"{d}/delfi_w{world}.tsv")
    counted = sorted(set(asked))
    cov = Cv.coverage("x", f"{d}/iv.bed", f"{d}/cov_w{world}.bed", normalize=True, scale_factor=1e6)
    # an interval on a contig the input lacks fails on the rank that owns it - and must fail on EVERY rank (the owner
    # with its own exception, the others with a RuntimeError naming it), not leave them in the gather
    region_reads = list(src.region_reads)
    try:
        Cv.coverage("x", f"{d}/iv_bad.bed", None)
        failed = "no error"
    except RuntimeError as e:
        failed = "other: " + str(e)
    except Exception as e:  # noqa: BLE001
        failed = "own: " + type(e).__name__
    sharding.finalize()
    q.put((rank, df.to_csv(), [tuple(c) for c in cov], counted, region_reads, failed))


def test_sharded_delfi_and_coverage_equal_single_process(tmp_path):
    import torch.multiprocessing as mp
    d = tmp_path
    rng = np.random.default_rng(3)
    (d / "cs.genome").write_text("".join(f"{c}\t{n}\n" for c, n in SIZES_P.items()))
    (d / "bins.txt").write_text("".join(f"{c}\t{a}\t{min(a + 9_999, n)}\n" for c, n in SIZES_P.items()
                                        for a in range(0, n, 10_000)))
    (d / "gaps.bed").write_text("".join(f"{c}\t0\t10000\ttelomere\n{c}\t{n // 20000 * 10000}\t{n // 20000 * 10000 + 30000}"
                                        f"\tcentromere\n{c}\t{n - 10000}\t{n}\ttelomere\n" for c, n in SIZES_P.items()))
    (d / "bl.bed").write_text("".join(f"{c}\t{int(a)}\t{int(a) + 900}\n" for c, n in SIZES_P.items()
                                      for a in rng.integers(0, n - 1000, 20)))
    iv = [f"{c}\t{int(a)}\t{int(a) + int(rng.integers(1, 4000))}\t{c}{k}\n" for c, n in SIZES_P.items()
          for k, a in enumerate(rng.integers(0, n - 4000, 400 if c == "b" else 30))]  # (b is half of the cost: wherever the shuffle puts it, the cut of two equal-cost runs falls into it)
    rng.shuffle(iv)
    (d / "iv.bed").write_text("".join(iv))
    (d / "iv_bad.bed").write_text("".join(iv[:40]) + "nope\t10\t500\tmissing\n" + "".join(iv[40:80]))
    ctx = mp.get_context("spawn")
    res = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_product_worker, args=(r, world, port, str(d), q)) for r in range(world)]
        for p in procs:
            p.start()
        res[world] = sorted(q.get(timeout=300) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    one = res[1][0]
    assert one[3] == sorted(SIZES_P) and len(one[2]) == 520 and one[1].count("\n") > 200
    for r in res[2]:
        assert r[1] == one[1] and r[2] == one[2]  # same frame, same coverage list on every rank
    a, b = set(res[2][0][3]), set(res[2][1][3])
    # the bins were cut into two runs of equal cost (sharding.split_counts): one contig is shared, the others counted by
    # exactly one rank
    assert a and b and len(a & b) == 1 and a | b == set(SIZES_P), (a, b)
    assert (d / "delfi_w2.tsv").read_text() == (d / "delfi_w1.tsv").read_text()
    assert (d / "cov_w2.bed").read_text() == (d / "cov_w1.bed").read_text()
    # coverage's intervals take the same kind of partition (sharding.IntervalPlan): one process reads no region; with two
    # ranks the contig the cut falls into is read as a region by both, each spanning only its own intervals
    assert one[4] == [] and one[5].startswith("own: ")
    assert sorted(r[5].split(":")[0] for r in res[2]) == ["other", "own"] and any("rank" in r[5] and "failed" in r[5] for r in res[2])
    r0, r1 = res[2][0][4], res[2][1][4]
    assert len(r0) == 1 and len(r1) == 1 and r0[0][0] == r1[0][0] and r0[0][1:] != r1[0][1:]
    assert r0[0][2] <= r1[0][1] + 4001 or r1[0][2] <= r0[0][1] + 4001  # (start-ordered shares: the two regions barely overlap)


def test_launch_ranks_starts_n_ranks_and_propagates_failure(tmp_path):
    from finaletoolkit_amd import sharding
    ok = tmp_path / "ok.py"
    ok.write_text(f"import os, sys\nsys.path.insert(0, {ROOT!r})\nfrom finaletoolkit_amd import sharding\n"
                  "r, w = sharding.init_from_env()\nassert sharding.allreduce_sum(r + 1) == w * (w + 1) // 2\n"
                  f"open({str(tmp_path)!r} + f'/seen{{r}}', 'w').write(os.environ['LOCAL_RANK'])\nsharding.finalize()\n")
    env_keep = {k: os.environ.pop(k, None) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    try:
        assert sharding.launch_ranks([sys.executable, str(ok)], 3, share_gpu=True) == 0
        assert sorted(p.name for p in tmp_path.glob("seen*")) == ["seen0", "seen1", "seen2"]
        bad = tmp_path / "bad.py"
        bad.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(600)\n")
        assert sharding.launch_ranks([sys.executable, str(bad)], 2, share_gpu=True) == 7  # rank 0 is ended, not waited for
        # without share_gpu the device count decides: this container has no GPU at all
        import torch
        if torch.cuda.device_count() < 2:
            assert sharding.launch_ranks([sys.executable, str(ok)], 2) == 2
    finally:
        for k, v in env_keep.items():
            if v is not None:
                os.environ[k] = v


# ---- the helpers behind the other sharded commands (multi_wps, cleavage, frag_length_*, motifs) --------------

def _helpers_worker(rank, world, port, d, q):
    sys.path.insert(0, ROOT)
    os.environ["FTK_SHARD_OVERHEAD_BASES"] = "0"  # (contigs of a few 10 kb: the shares are cut by the intervals' span alone)
    from finaletoolkit_amd import sharding
    from finaletoolkit_amd.frag import _runs
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          FTK_DIST_BACKEND="gloo")
        sharding.init_from_env()
    # payload gather: item k made by rank k % world, sizes from 0 to a few hundred KB
    owner = [k % world for k in range(7)]
    local = {k: bytes([k]) * (k * 50_021) for k in range(7) if owner[k] == rank}
    got = sharding.gather_payloads(local, owner)
    assert (got is None) == (rank != 0)
    if rank == 0:
        assert [len(g) for g in got] == [k * 50_021 for k in range(7)] and all(set(g) <= {k} for k, g in enumerate(got))
    # float rows travel as bit patterns
    names = ["x", "y", "z"]
    own = {"x": 0, "y": world - 1, "z": 0}
    rows = {c: np.array([[np.pi * (i + 1), -0.0, np.nan, 1e-310, float(i), 3.0, 2.0 ** 60]], np.float64).repeat(i + 2, 0)
            for i, c in enumerate(names)}
    full = sharding.gather_float_rows({c: v for c, v in rows.items() if own[c] == rank}, names,
                                      {c: len(rows[c]) for c in names}, own, 7)
    for c in names:
        assert full[c].tobytes() == rows[c].tobytes()
    assert sharding.allgather_object({"r": rank}) == [{"r": r} for r in range(world)]
    # unit rows: the bins of three contigs cut into `world` runs of equal cost; every rank makes the rows of its units
    # (a function of the unit), one all-gather hands every rank every contig's rows in order
    counts = {"x": 37, "y": 5, "z": 22}
    units = sharding.split_counts(counts, world, overhead=2)
    assert sorted((c, i) for _, c, i0, i1 in units for i in range(i0, i1)) == sorted((c, i) for c, n in counts.items() for i in range(n))
    assert [u[0] for u in units] == sorted(u[0] for u in units) and {u[0] for u in units} == set(range(world))
    n_rows = {(c, i0, i1): (i1 - i0) - (i1 - i0) // 3 for _, c, i0, i1 in units}  # (some bins are not live)
    make = lambda c, i0, i1: (np.arange(n_rows[(c, i0, i1)] * 4, dtype=np.int64).reshape(-1, 4) + 1000 * i0 + ord(c))  # noqa: E731
    got_rows = sharding.gather_unit_rows({(c, i0, i1): make(c, i0, i1) for r, c, i0, i1 in units if r == rank}, units, n_rows, 4)
    for c in counts:
        want = np.concatenate([make(cc, i0, i1) for _, cc, i0, i1 in units if cc == c])
        assert np.array_equal(got_rows[c], want), c
    # agree(): one failing rank makes every rank raise
    try:
        sharding.agree(ValueError("boom") if rank == world - 1 else None)
        raised = None
    except ValueError as e:
        raised = "own:" + str(e)
    except RuntimeError as e:
        raised = "other:" + str(e)
    # per-base run outputs: every rank "scores" its contigs, rank 0 writes
    header = [("a", 50_000), ("b", 40_000), ("c", 30_000)]
    contigs = ["a"] * 3 + ["b"] * 2 + ["c"] * 4 + ["a"]  # the last run comes back to `a`: skipped in the bigWig
    starts = [100, 5000, 20_000, 10, 9000, 0, 300, 7000, 29_000, 40_000]
    stops = [1100, 5000, 21_500, 4010, 9100, 200, 1300, 7001, 30_000, 41_000]
    asked, tables = [], []

    class Src:  # what write_per_base_runs asks a FragSource for
        def require(self, c):
            tables.append((c, None, None))
            return c

        def require_region(self, c, lo, hi):
            tables.append((c, int(lo), int(hi)))
            return f"{c}@{lo}-{hi}"

        def release_region(self, key):
            pass

    def compute(key, c, st, sp):
        asked.append((c, tuple(st)))
        assert key == c or key.startswith(c + "@")
        if "@" in key:  # a region table: it must span every interval it is asked about
            lo, hi = (int(x) for x in key.split("@")[1].split("-"))
            assert all(lo <= a and b <= hi for a, b in zip(st, sp) if b > a), (key, st, sp)
        offs = np.concatenate([[0], np.cumsum([b - a for a, b in zip(st, sp)])]).astype(np.int64)
        vals = np.concatenate([np.arange(a, b, dtype=np.int64) * (ord(c) - 96) - 7 for a, b in zip(st, sp)] or
                              [np.zeros(0, np.int64)])
        return vals, offs

    _runs.UNIT_BASES = 1200  # (units of a few intervals: the ranks' shares are cut inside contigs)
    assert len(_runs.split_into_units(_runs.group_runs(contigs), starts, stops)) == 7
    _runs.write_per_base_runs(f"{d}/w{world}.bw", "bw", header, contigs, starts, stops, compute, Src(), 5)
    _runs.write_per_base_runs(f"{d}/w{world}.bed.gz", "bedgraph.gz", header, contigs, starts, stops,
                              lambda k, c, st, sp: (compute(k, c, st, sp)[0].astype(np.float64) / 3.0, compute(k, c, st, sp)[1]),
                              Src(), 5)
    sharding.finalize()
    q.put((rank, raised, sorted(set(asked)), sorted(set(tables), key=str)))


def test_payload_gather_float_rows_agree_and_run_outputs(tmp_path):
    import gzip
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for world in (1, 2, 3):
        q = ctx.Queue()
        port = _free_port()
        procs = [ctx.Process(target=_helpers_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
        for p in procs:
            p.start()
        res[world] = sorted(q.get(timeout=300) for _ in procs)
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    assert res[1][0][1] == "own:boom"
    assert [r[1][:6] for r in res[2]] == ["other:", "own:bo"] and "rank 1 failed: ValueError: boom" in res[2][0][1]
    for world in (2, 3):
        owned = [set(r[2]) for r in res[world]]
        # every unit scored by exactly one rank (the bigWig skips the run that comes back to `a`, the bedGraph does not)
        assert set().union(*owned) == set(res[1][0][2]) and sum(map(len, owned)) == len(res[1][0][2])
        # a rank that scores only part of a contig's units asked for a REGION of it, the single process never did
        assert any(t[1] is not None for r in res[world] for t in r[3]) and all(t[1] is None for t in res[1][0][3])
        assert (tmp_path / f"w{world}.bw").read_bytes() == (tmp_path / "w1.bw").read_bytes()
        assert (tmp_path / f"w{world}.bed.gz").read_bytes() == (tmp_path / "w1.bed.gz").read_bytes()
    from finaletoolkit_amd.bigwig import BigWigFile
    with BigWigFile(tmp_path / "w1.bw") as bw:
        s, e, v = bw.intervals("a", 0, 50_000)
        assert len(s) == 1000 + 1500 and s[0] == 100 and v[0] == 93.0  # the run that came back to `a` is not there
        s, e, v = bw.intervals("c", 0, 30_000)
        assert len(s) == 200 + 1000 + 1 + 1000 and v[0] == -7.0
    text = gzip.open(tmp_path / "w1.bed.gz", "rt").read().splitlines()
    assert len(text) == 1000 + 1500 + 4000 + 100 + 200 + 1000 + 1 + 1000 + 1000
    assert text[0] == f"a\t100\t101\t{93 / 3.0!r}" and text[-1] == f"a\t40999\t41000\t{(40999 - 7) / 3.0!r}"


def test_cli_refuses_gpus_for_unsharded_commands(tmp_path):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = tmp_path / "gaps.bed"
    r = subprocess.run([sys.executable, "-m", "finaletoolkit_amd.cli", "--gpus", "2", "gap-bed", "hg19", str(out)],
                       cwd=ROOT, env=env, capture_output=True, text=True)
    assert r.returncode == 2 and "does not shard" in r.stderr and not out.exists()
    r = subprocess.run([sys.executable, "-m", "finaletoolkit_amd.cli", "gap-bed", "hg19", str(out)], cwd=ROOT,
                       env=dict(env, WORLD_SIZE="2", RANK="1", LOCAL_RANK="1"), capture_output=True, text=True)
    assert r.returncode == 2 and not out.exists()
    r = subprocess.run([sys.executable, "-m", "finaletoolkit_amd.cli", "gap-bed", "hg19", str(out)], cwd=ROOT, env=env,
                       capture_output=True, text=True)
    assert r.returncode == 0 and out.exists()


# ---- world 8 (the node the product is sized for), gloo on CPU ---------------------------------------------------
def _rows_of(contig, i0, i1):
    """Deterministic stand-in for a unit's counted rows: the LIVE items of [i0, i1) (every third one is not), four
    int64 columns derived from the contig and the item - any rank can compute any unit's rows, so every rank can check
    the whole gathered table."""
    live = [i for i in range(i0, i1) if i % 3 != 1]
    seed = sum(ord(ch) for ch in contig)
    return np.array([[seed * 1_000_003 + i, i * i, -i, (seed << 40) + i] for i in live], np.int64).reshape(-1, 4)


def _world8_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from finaletoolkit_amd import sharding, synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    report = {}
    try:
        # (1) the product's partition on b37's 100 kb tiling, and (2) a tiny table that leaves ranks empty-handed
        b37 = {c: -(-n // 100_000) for c, n in synth.B37_SIZES.items()}
        tiny = {"p": 3, "q": 0, "r": 2}
        for label, counts in (("b37", b37), ("tiny", tiny)):
            units = sharding.split_counts(counts, world)
            n_rows = {(c, i0, i1): len(_rows_of(c, i0, i1)) for _, c, i0, i1 in units}
            local = {(c, i0, i1): _rows_of(c, i0, i1) for r, c, i0, i1 in units if r == rank}
            got = sharding.gather_unit_rows(local, units, n_rows, 4)
            want = {c: _rows_of(c, 0, n) for c, n in counts.items() if n > 0}
            report[label] = (sorted(got) == sorted(want) and all(np.array_equal(got[c], want[c]) for c in want),
                             len(local), int(sum(len(v) for v in local.values())))
        # (3) byte payloads of numbered items to rank 0: unequal sizes, an empty one, ranks that own none
        owner = [(7 * k + 3) % world if k % 5 else 2 for k in range(23)]  # rank 2 owns many, some ranks none
        owner = [o if o not in (4, 6) else 1 for o in owner]              # ranks 4 and 6: empty-handed
        mine = {k: (bytes([k]) * (0 if k == 11 else 1000 * k + 17)) for k in range(23) if owner[k] == rank}
        out = sharding.gather_payloads(mine, owner)
        if rank == 0:
            report["payloads"] = out == [bytes([k]) * (0 if k == 11 else 1000 * k + 17) for k in range(23)]
        else:
            report["payloads"] = out is None
        # (4) float rows with LPT owners, (5) the scalar all-reduce, (6) agreement on an error raised by ONE rank
        names = list(synth.B37_SIZES)
        lpt = sharding.lpt_assign({c: float(b37[c]) for c in names}, world)
        n_f = {c: 1 + b37[c] % 7 for c in names}
        fl = {c: np.full((n_f[c], 2), 0.1 * names.index(c)) + np.arange(2 * n_f[c]).reshape(-1, 2) / 3.0 for c in names}
        gf = sharding.gather_float_rows({c: fl[c] for c in names if lpt[c] == rank}, names, n_f, lpt, 2)
        report["floats"] = all(np.array_equal(gf[c], fl[c]) for c in names)
        report["sum"] = sharding.allreduce_sum(rank + 1) == world * (world + 1) // 2
        try:
            sharding.agree(ValueError("rank five's share failed") if rank == 5 else None)
            report["agree"] = False
        except ValueError:
            report["agree"] = rank == 5
        except RuntimeError as e:
            report["agree"] = rank != 5 and "rank 5 failed" in str(e)
        dist.barrier()
    finally:
        dist.destroy_process_group()
    q.put((rank, report))


def test_world_8_partition_and_gathers():
    """`split_counts` / `gather_unit_rows` / `gather_payloads` / `gather_float_rows` / `agree` at the world size the
    product is sized for (one node of eight MI355X), over gloo: a rank owning pieces of three contigs, ranks without a
    unit, unequal shard lengths - so that the first real 8-GPU run can only fail on hardware grounds."""
    import torch.multiprocessing as mp
    from finaletoolkit_amd import sharding, synth
    world = 8
    b37 = {c: -(-n // 100_000) for c, n in synth.B37_SIZES.items()}
    units = sharding.split_counts(b37, world)
    # the partition itself: every item once, in order; equal cost; a rank with three contigs' pieces; b37 balance
    for c, n in b37.items():
        cuts = [(i0, i1) for _, cc, i0, i1 in units if cc == c]
        assert cuts[0][0] == 0 and cuts[-1][1] == n and all(a[1] == b[0] for a, b in zip(cuts, cuts[1:]))
    per_rank = {r: [u for u in units if u[0] == r] for r in range(world)}
    assert max(len({u[1] for u in us}) for us in per_rank.values()) >= 3
    assert [u[0] for u in units] == sorted(u[0] for u in units)          # ranks take consecutive runs of the genome
    loads = [sum(i1 - i0 for _, _, i0, i1 in us) for us in per_rank.values()]
    assert (sum(loads) / world) / max(loads) >= 0.98
    tiny_units = sharding.split_counts({"p": 3, "q": 0, "r": 2}, world)
    assert len({u[0] for u in tiny_units}) < world                        # some ranks hold nothing
    assert sorted((c, i) for _, c, i0, i1 in tiny_units for i in range(i0, i1)) == [("p", 0), ("p", 1), ("p", 2), ("r", 0), ("r", 1)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=600) for _ in procs)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(res) == list(range(world))
    for rank, rep in res.items():
        assert rep["b37"][0] and rep["tiny"][0], (rank, rep)
        assert rep["payloads"] and rep["floats"] and rep["sum"] and rep["agree"], (rank, rep)
    assert sum(rep["b37"][1] for rep in res.values()) == len(units)
    assert min(rep["tiny"][1] for rep in res.values()) == 0              # an empty-handed rank went through the gather
    assert len({rep["b37"][2] for rep in res.values()}) > 1              # unequal shard lengths
