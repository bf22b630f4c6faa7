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
