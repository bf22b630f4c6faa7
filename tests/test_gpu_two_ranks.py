"""GPU: the multi-rank product path with HIP compute.  Two rank processes share GPU 0 (exchange over gloo
host copies, what a 1-GPU box allows); each decodes and counts only ITS contigs on the device, the bin vectors
meet in one all-gather, and every rank must return exactly what a single process returns
(reference fan-out: frag/_delfi.py:289-300, frag/_coverage.py:212-248).  Also: ``bench.py --gpus N`` starts N
ranks or fails loudly -- it never benchmarks fewer GPUs than asked for."""
import json
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu
ROOT = H.ROOT

WORKER = r"""
import os, pickle, sys, warnings
sys.path.insert(0, {root!r})
from finaletoolkit_amd import frag, sharding
rank, world = sharding.init_from_env()
d = {tmp!r}
with warnings.catch_warnings():
    warnings.simplefilter("ignore")
    df = frag.delfi(d + "/g.frag.gz", d + "/cs.genome", d + "/bins.txt", d + "/ref.fa", blacklist_file=d + "/bl.bed",
                    gap_file=d + "/gaps.bed", no_gc_correct=True, remove_nocov=False, merge_bins=False,
                    output_file=d + f"/delfi_w{{world}}.tsv")
    merged = frag.delfi(d + "/g.frag.gz", d + "/cs.genome", d + "/bins.txt", d + "/ref.fa", gap_file=d + "/gaps.bed",
                        no_gc_correct=True, remove_nocov=False, merge_bins=True)
cov = frag.coverage(d + "/g.frag.gz", d + "/iv.bed", d + f"/cov_w{{world}}.bed", normalize=True, scale_factor=1e6)
raw = frag.coverage(d + "/g.frag.gz", d + "/iv.bed", None, intersect_policy="any", min_length=100, max_length=220)
from finaletoolkit_amd.source import get_engine
loaded = sorted(k.split(":", 1)[1] for k in get_engine().contigs)
pickle.dump(dict(rank=rank, world=world, delfi=df, merged=merged, cov=[tuple(c) for c in cov],
                 raw=[tuple(c) for c in raw], loaded=loaded), open(d + f"/out_w{{world}}_r{{rank}}.pkl", "wb"))
sharding.finalize()
"""


@pytest.fixture(scope="module")
def dataset(tmp_path_factory):
    from finaletoolkit_amd import bgzf, synth
    d = tmp_path_factory.mktemp("two_ranks")
    sizes = {"c1": 900_000, "c2": 700_000, "c3": 420_000, "c4": 350_000, "c5": 130_000}
    rows = []
    for i, (c, n) in enumerate(sizes.items()):
        s, e, q, st = synth.synth_contig(n, depth=6.0, seed=900 + i)
        rows.append((c, s, e, q, st))
    bgzf.write_frag_gz(d / "g.frag.gz", rows, level=1, with_index=True)
    (d / "cs.genome").write_text("".join(f"{c}\t{n}\n" for c, n in sizes.items()))
    (d / "bins.txt").write_text("".join(f"{c}\t{a}\t{min(a + 9_999, n)}\n" for c, n in sizes.items()
                                        for a in range(0, n, 10_000)))
    seqs = {}
    rng = np.random.default_rng(4)
    for c, n in sizes.items():
        seqs[c] = np.frombuffer(b"ACGTNacgt", np.uint8)[rng.integers(0, 9, n)].tobytes().decode()
    H.write_fasta(d / "ref.fa", seqs)
    (d / "gaps.bed").write_text("".join(
        f"{c}\t0\t10000\ttelomere\n{c}\t{n // 2 // 10000 * 10000}\t{n // 2 // 10000 * 10000 + 30000}\tcentromere\n"
        f"{c}\t{n - 10000}\t{n}\ttelomere\n" for c, n in sizes.items()))
    bl = []
    for c, n in sizes.items():
        for a in rng.integers(0, n - 3000, 25):
            bl.append(f"{c}\t{int(a)}\t{int(a) + int(rng.integers(100, 2500))}\n")
    (d / "bl.bed").write_text("".join(bl))
    iv = []
    for c, n in sizes.items():
        for k, a in enumerate(rng.integers(0, n - 5000, 40)):
            iv.append(f"{c}\t{int(a)}\t{int(a) + int(rng.integers(1, 5000))}\t{c}_{k}\n")
    rng.shuffle(iv)  # interval order is file order, not contig order
    (d / "iv.bed").write_text("".join(iv))
    (d / "worker.py").write_text(WORKER.format(root=ROOT, tmp=str(d)))
    return d


def _run_world(d, world):
    from finaletoolkit_amd import sharding
    env_keep = {k: os.environ.get(k) for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for k in env_keep:
        os.environ.pop(k, None)
    try:
        if world == 1:
            rc = subprocess.run([sys.executable, str(d / "worker.py")], cwd=ROOT).returncode
        else:
            rc = sharding.launch_ranks([sys.executable, str(d / "worker.py")], world, share_gpu=True)
    finally:
        for k, v in env_keep.items():
            if v is not None:
                os.environ[k] = v
    assert rc == 0
    return [pickle.load(open(d / f"out_w{world}_r{r}.pkl", "rb")) for r in range(world)]


def test_two_ranks_equal_one_process(dataset):
    d = dataset
    one = _run_world(d, 1)[0]
    two = _run_world(d, 2)
    assert [r["world"] for r in two] == [2, 2]
    for r in two:  # every rank returns the single-process result, bit for bit
        assert r["delfi"].equals(one["delfi"]) and r["merged"].equals(one["merged"])
        assert r["cov"] == one["cov"] and r["raw"] == one["raw"]
    assert one["delfi"]["num_frags"].sum() > 10_000 and len(one["cov"]) == 200
    # and rank 0 alone wrote the files, identical to the single-process ones
    assert open(d / "delfi_w2.tsv").read() == open(d / "delfi_w1.tsv").read()
    assert open(d / "cov_w2.bed").read() == open(d / "cov_w1.bed").read()
    # the work really was dealt out: each rank decoded only its own contigs, together all of them
    a, b = set(two[0]["loaded"]), set(two[1]["loaded"])
    assert a and b and not (a & b) and a | b == set(one["loaded"]) == {"c1", "c2", "c3", "c4", "c5"}


def test_cli_gpus_2_writes_the_single_process_file(dataset, tmp_path):
    d = dataset
    outs = {}
    for world in (1, 2):
        out = tmp_path / f"cli_w{world}.tsv"
        env = dict(os.environ, FTK_SHARE_GPU="1")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, "-m", "finaletoolkit_amd.cli", "--gpus", str(world), "delfi",
                            str(d / "g.frag.gz"), str(d / "cs.genome"), str(d / "ref.fa"), str(d / "bins.txt"),
                            "-b", str(d / "bl.bed"), "-g", str(d / "gaps.bed"), "--no-gc-correct", "--no-remove-nocov", "--no-merge-bins",
                            "-o", str(out)],
                           cwd=ROOT, env=env, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[world] = open(out).read()
    assert outs[1] == outs[2] and outs[1].count("\n") > 5


def test_bench_gpus_n_never_runs_fewer_ranks(tmp_path):
    """`python bench.py --gpus N` on a box with fewer than N devices exits non-zero with a clear message; a
    launcher that started a different number of ranks than --gpus is refused too."""
    import torch
    have = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(have + 1), "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       env=env, capture_output=True, text=True)
    assert r.returncode != 0 and f"needs {have + 1} visible" in r.stderr and not r.stdout.strip()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "4", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"), capture_output=True, text=True)
    assert r.returncode != 0 and "must agree" in r.stderr


def test_bench_two_ranks_share_the_gpu():
    """bench.py's own N-rank path (unit split, per-rank launches, all-gather of the DELFI vector) with two ranks
    on GPU 0: one JSON line from rank 0, n_gpus 2, every check true."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(FTK_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--steps", "2", "--warmup", "1", "--contigs",
                        "20,21,22", "--no-cpu-baseline", "--no-end-to-end"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["checks"] and all(line["checks"].values()), line["checks"]
    assert "gloo" in line["exchange"] and "2 ranks" in line["exchange"]
