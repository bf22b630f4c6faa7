"""
Multi-GPU sharding of the hot path: one process per GPU, contigs assigned to
ranks by longest-processing-time greedy (every window / bin / base depends only
on the fragments of its own contig, so no data-path collective is needed), and
one all-gather of the fixed-size per-bin vectors so every rank ends up with the
whole-genome DELFI / coverage vector in contig order.  The transport is
``comm.py``: the library's own RCCL communicator (``ftk_comm_*``, include/ftk.h;
no torch in the process) on the GPU box, ``torch.distributed``/gloo in CPU tests
and when several ranks share one GPU; payloads are a few hundred KB, i.e.
latency-bound.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def launch_ranks(cmd: Sequence[str], n: int, share_gpu: bool = False) -> int:
    """Start ``n`` fresh rank processes of ``cmd`` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their
    environment, one GPU each) and wait for them; rank 0 inherits stdout, the other ranks' stdout is dropped.
    The calling process must not have initialised the GPU and does not here
    (``torch.cuda.device_count()`` does not).  Returns the exit status: 2 when fewer than ``n`` devices are
    visible, else the first failing rank's (a failed rank ends the others, which would wait in a collective
    forever).  ``share_gpu`` (tests on a 1-GPU box): every rank on GPU 0, exchange through gloo."""
    import socket
    import subprocess
    import sys
    import time
    if not share_gpu:
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write(f"--gpus {n} needs {n} visible MI355X devices, found {have} "
                             f"(one rank per GPU; no fallback to fewer)\n")
            return 2
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        env["MASTER_PORT"] = str(s.getsockname()[1])
        s.close()
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if share_gpu:
        env.setdefault("FTK_DIST_BACKEND", "gloo")
    # The ranks of THIS launch meet in a file of their own: a fresh directory only this user can enter, and a nonce the
    # file must carry - nothing a killed earlier launch left behind, and nothing another user put there, can be taken
    # for it (ftk_comm_create; comm.id_file).
    import secrets
    import shutil
    import tempfile
    meet = None
    if "FTK_COMM_ID_FILE" not in env:
        meet = tempfile.mkdtemp(prefix="ftk_comm_")  # (mode 0700)
        env["FTK_COMM_ID_FILE"] = os.path.join(meet, "id")
    env.setdefault("FTK_COMM_NONCE", secrets.token_hex(16))
    procs = []
    rc = 0
    try:
        for r in range(n):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(0 if share_gpu else r))
            procs.append(subprocess.Popen(list(cmd), env=e, stdout=None if r == 0 else subprocess.DEVNULL))
        live = list(procs)
        while live:
            time.sleep(0.05)
            for p in list(live):
                if p.poll() is not None:
                    live.remove(p)
                    rc = rc or p.returncode
            if rc:
                for p in live:
                    p.kill()
                    p.wait()
                break
    finally:
        if meet:
            shutil.rmtree(meet, ignore_errors=True)
    return rc


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int]:
    """Join the job's group when this process was started as one rank of several (``torchrun``, or
    ``python -m finaletoolkit_amd.cli --gpus N``: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the
    environment): the library's RCCL communicator on GPU ``LOCAL_RANK`` (``comm.RcclGroup``, no torch) unless
    ``FTK_DIST_BACKEND`` / ``backend`` says ``gloo`` (CPU tests; several ranks sharing one GPU) or ``nccl``
    (``torch.distributed``).  Returns ``(rank, world)``; a plain single process returns ``(0, 1)`` and starts
    nothing.  This is the counterpart of the reference's ``Pool(workers)`` (frag/_delfi.py:289,
    frag/_coverage.py:212): one rank per GPU."""
    from . import comm
    g = comm.join(backend)
    return g.rank, g.world


def finalize():
    """Leave the group (end of a multi-rank command)."""
    import sys
    from . import comm
    if comm._GROUP is not None:
        comm.leave()
        return
    if "torch.distributed" not in sys.modules:
        return
    try:
        import torch.distributed as dist
    except ImportError:  # pragma: no cover
        return
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def rank_world(group=None) -> Tuple[int, int]:
    """``(rank, world)`` of the group this process exchanges in, ``(0, 1)`` without one.  Never imports torch: a
    process that has not imported ``torch.distributed`` cannot hold one of its groups (seconds on a fresh box, and
    every writer asks)."""
    from . import comm
    g = comm.current(group)
    return g.rank, g.world


def lpt_assign(weights: Dict[str, float], n_ranks: int) -> Dict[str, int]:
    """Longest-processing-time greedy: heaviest contig first onto the least loaded rank."""
    loads = [0.0] * n_ranks
    owner = {}
    for name in sorted(weights, key=lambda k: (-weights[k], k)):
        r = loads.index(min(loads))
        owner[name] = r
        loads[r] += weights[name]
    return owner


def split_counts(counts: Dict[str, int], n_ranks: int, overhead: int = 50):
    """THE partition of the product and the bench: contigs in the given order, each with ``counts[c]`` items (the
    windows of a tiling, the rows of a bins file), laid end to end and cut into ``n_ranks`` runs of equal COST, so a
    rank owns whole contigs plus at most two partial ones (whole-contig LPT caps 8 GPUs at 0.96 of ideal on b37:
    chr1 alone is 8 % of the genome).  Cost = items + ``overhead`` per contig: every unit pays launch ramps, an index
    seek and a first block worth about 50 windows of streaming (measured on simulated ranks: a rank of six small
    contigs ran 6 % longer than one of two large ones with equal bases).  Returns ``[(rank, contig, i0, i1), ...]``
    in the given order: items ``[i0, i1)`` of ``contig`` belong to ``rank``; every item belongs to exactly one unit."""
    if n_ranks <= 1:
        return [(0, c, 0, int(n)) for c, n in counts.items() if int(n) > 0]
    n_it = {c: int(n) for c, n in counts.items()}
    ov = max(0, int(overhead))
    total = sum(n_it.values()) + ov * len(n_it)
    units = []
    done = 0.0  # cost before the current contig
    for c in counts:
        done += ov  # the contig's fixed cost sits in front of its first item
        w0 = 0
        while w0 < n_it[c]:
            r = min(n_ranks - 1, int((done + w0) * n_ranks // total))
            # first item of this contig whose cost position belongs to the next rank
            nxt = -(-(r + 1) * total // n_ranks) if r + 1 < n_ranks else total + n_it[c]
            w1 = min(n_it[c], max(w0 + 1, int(nxt - done)))
            units.append((r, c, w0, w1))
            w0 = w1
        done += n_it[c]
    return units


def shard_overhead_bases() -> int:
    """Fixed cost of entering a contig, in bases of decoded rows: what ``split_counts`` charges as 50 windows of
    100 kb (an index seek, a first block, launch ramps - measured on simulated ranks, see there).
    ``FTK_SHARD_OVERHEAD_BASES`` overrides it (tests with contigs far smaller than that set 0, so that the cut is
    taken by the intervals' span alone)."""
    v = os.environ.get("FTK_SHARD_OVERHEAD_BASES")
    return max(0, int(v)) if v not in (None, "") else 5_000_000


def split_weighted(weights: Dict[str, Sequence[float]], n_ranks: int, overhead: float):
    """``split_counts`` with a cost per item: contigs in the given order, contig ``c``'s items costing
    ``weights[c][i]`` each plus ``overhead`` per contig, cut into ``n_ranks`` consecutive runs of equal cost.  An item
    belongs to the rank its cost interval STARTS in.  Returns ``[(rank, contig, i0, i1), ...]`` like ``split_counts``."""
    items = {c: np.asarray(w, dtype=np.float64) for c, w in weights.items()}
    if n_ranks <= 1:
        return [(0, c, 0, len(w)) for c, w in items.items() if len(w) > 0]
    total = float(sum(w.sum() for w in items.values()) + overhead * len(items))
    units, done = [], 0.0
    for c, w in items.items():
        done += overhead
        if len(w):
            at = done + np.concatenate(([0.0], np.cumsum(w)[:-1]))  # where each item's cost starts
            rank = np.minimum(n_ranks - 1, (at * n_ranks / max(total, 1e-300)).astype(np.int64))
            cuts = np.flatnonzero(np.diff(rank)) + 1
            bounds = np.concatenate(([0], cuts, [len(w)]))
            for a, b in zip(bounds[:-1], bounds[1:]):
                units.append((int(rank[a]), c, int(a), int(b)))
        done += float(w.sum())
    return units


def split_units(sizes: Dict[str, int], n_ranks: int, window: int, unit_overhead_windows: int = 50):
    """``split_counts`` for a tiling of ``window`` bases (``bench.py``'s steps): ``[(rank, contig, start, stop), ...]``
    in genome order, ``start`` a multiple of ``window``.  A unit needs the contig's fragments starting in
    ``[start - halo, stop + halo)`` (``unit_halo``) and produces exactly the windows / bases of its own range:
    units never exchange data."""
    if n_ranks <= 1:
        return [(0, c, 0, int(n)) for c, n in sizes.items()]
    n_win = {c: -(-int(n) // window) for c, n in sizes.items()}
    return [(r, c, w0 * window, min(w1 * window, int(sizes[c])))
            for r, c, w0, w1 in split_counts(n_win, n_ranks, unit_overhead_windows)]


def unit_halo(max_fragment_len: int, wps_window: int) -> int:
    """Halo (bp) of a unit: a fragment can count in a window (midpoint policy, DELFI) or touch a WPS
    position of the unit only if it starts within ``max_fragment_len + wps_window`` of it."""
    return int(max_fragment_len) + int(wps_window)


def shard_contigs(names: Sequence[str], weights: Dict[str, float], rank: int, world: int) -> List[str]:
    owner = lpt_assign({n: weights[n] for n in names}, world)
    return [n for n in names if owner[n] == rank]


def gather_bin_vectors(local: Dict[str, np.ndarray], names: Sequence[str], n_bins: Dict[str, int],
                       weights: Dict[str, float], group=None, device=None, k: Optional[int] = None,
                       owner: Optional[Dict[str, int]] = None) -> Dict[str, np.ndarray]:
    """All-gather per-contig integer vectors (shape [n_bins[c], k]) so every rank
    holds all contigs.  ``local`` has this rank's contigs; ``n_bins`` the row
    count of EVERY contig (known from the bin file on all ranks)."""
    from . import comm
    g = comm.current(group)
    if g.world == 1:  # (decided without importing torch: seconds on every single-process call)
        return dict(local)
    world, rank = g.world, g.rank
    if owner is None:
        owner = lpt_assign({n: weights[n] for n in names}, world)
    if k is None:
        k = next((v.shape[1] for v in local.values()), None)
        k = next(x for x in g.all_gather_object(k) if x is not None)
    rows = [sum(n_bins[n] for n in names if owner[n] == r) for r in range(world)]
    pad = max(max(rows), 1)  # RCCL does not take empty buffers
    send = np.zeros((pad, k), np.int64)
    off = 0
    for n in names:
        if owner[n] == rank:
            send[off:off + n_bins[n]] = np.ascontiguousarray(local[n], dtype=np.int64)
            off += n_bins[n]
    recv = g.all_gather_i64(send).reshape(world, pad, k)
    out = {}
    offs = [0] * world
    for n in names:
        r = owner[n]
        out[n] = recv[r, offs[r]:offs[r] + n_bins[n]].copy()
        offs[r] += n_bins[n]
    return out


def gather_unit_rows(local: Dict[tuple, np.ndarray], units, n_rows: Dict[tuple, int], k: int, group=None,
                     device=None) -> Dict[str, np.ndarray]:
    """All-gather the integer rows of ``split_counts`` units: ``local[(contig, i0, i1)]`` holds this rank's units
    (shape ``[n_rows[unit], k]``; every rank knows every unit's row count), the result maps each contig to its units'
    rows concatenated in unit order - on every rank.  One collective, like ``gather_bin_vectors``."""
    from . import comm
    g = comm.current(group)
    keys = [(c, i0, i1) for _, c, i0, i1 in units]
    if g.world == 1:  # (decided without importing torch)
        got = {key: np.asarray(local[key], dtype=np.int64).reshape(-1, k) for key in keys}
    else:
        world, rank = g.world, g.rank
        rows = [sum(n_rows[(c, i0, i1)] for r, c, i0, i1 in units if r == q) for q in range(world)]
        pad = max(max(rows), 1)  # RCCL does not take empty buffers
        send = np.zeros((pad, k), np.int64)
        off = 0
        for r, c, i0, i1 in units:
            if r == rank:
                n = n_rows[(c, i0, i1)]
                send[off:off + n] = np.ascontiguousarray(local[(c, i0, i1)], dtype=np.int64).reshape(n, k)
                off += n
        recv = g.all_gather_i64(send).reshape(world, pad, k)
        offs = [0] * world
        got = {}
        for r, c, i0, i1 in units:
            n = n_rows[(c, i0, i1)]
            got[(c, i0, i1)] = recv[r, offs[r]:offs[r] + n].copy()
            offs[r] += n
    out: Dict[str, list] = {}
    for key in keys:
        out.setdefault(key[0], []).append(got[key])
    return {c: np.concatenate(parts, axis=0) if parts else np.zeros((0, k), np.int64) for c, parts in out.items()}


class IntervalPlan:
    """THE partition of every interval-driven command (``coverage``, ``frag_length_intervals``, the motif drivers; the
    reference fans their intervals out over ``Pool(workers)``: frag/_coverage.py:212-248, frag/_frag_length.py:571-593,
    frag/_motif_common.py:635-685): a contig's intervals in start order, all contigs laid end to end in order of first
    appearance and cut into equal-cost consecutive runs (``split_weighted``: the partition ``frag.delfi`` and the bench
    use, ``split_counts``, with the BASES an interval adds to the span its share must decode as its cost - a region
    read decodes every row between its first and last interval - plus the fixed cost of entering a contig), so a rank
    owns whole contigs plus at most two partial ones.  A partial share is answered from a REGION of the
    contig (``unit_key``: the rows / records between its first interval's start and its last one's stop, read through
    the index), a whole one from the contig.  Results travel as fixed-width integer rows in ONE all-gather
    (``gather_unit_rows``) and come back in the input order of the intervals - on every rank."""

    def __init__(self, contigs: Sequence[str], starts: Sequence[int], stops: Sequence[int], group=None):
        self.rank, self.world = rank_world(group)
        self.group = group
        self.n = len(contigs)
        starts = np.asarray(starts, dtype=np.int64)
        stops = np.asarray(stops, dtype=np.int64)
        by: Dict[str, list] = {}
        for i, c in enumerate(contigs):
            by.setdefault(c, []).append(i)
        # a contig's intervals in start order (stable): consecutive shares are then compact regions of the contig
        self.order = {c: np.asarray(idx, dtype=np.int64)[np.argsort(starts[idx], kind="stable")] for c, idx in by.items()}
        self.starts, self.stops = starts, stops
        cost = {}
        for c, idx in self.order.items():  # an interval's cost: the bases up to the next one's start (the last: its own)
            st = starts[idx]
            nxt = np.concatenate((st[1:], [max(int(stops[idx].max()), int(st[-1]) + 1)]))
            cost[c] = np.maximum(nxt - st, 1)
        self.units = split_weighted(cost, self.world, shard_overhead_bases())
        self.mine = [(c, i0, i1) for r, c, i0, i1 in self.units if r == self.rank]

    def intervals(self, unit) -> np.ndarray:
        """Input positions of the unit's intervals (start order)."""
        c, i0, i1 = unit
        return self.order[c][i0:i1]

    def is_whole(self, unit) -> bool:
        c, i0, i1 = unit
        return i0 == 0 and i1 == len(self.order[c])

    def extent(self, unit):
        idx = self.intervals(unit)
        return int(self.starts[idx].min()), int(self.stops[idx].max())

    def unit_key(self, src, unit, pad: int = 1) -> str:
        """Engine key of a table that answers every query of the unit: the contig (a whole share, one process, a file
        that cannot be entered inside a contig) or the region its intervals span, ``pad`` bases either side."""
        c = unit[0]
        if self.world == 1 or self.is_whole(unit):
            return src.require(c)
        lo, hi = self.extent(unit)
        if hi <= lo:
            return src.require(c)
        return src.require_region(c, max(0, lo - pad), hi + pad)

    def release(self, src, key: str) -> None:
        if hasattr(src, "release_region"):
            src.release_region(key)  # (a no-op for a whole contig's key)

    def gather(self, local: Dict[tuple, np.ndarray], k: int, dtype=np.int64) -> np.ndarray:
        """``local[unit]``: ``[len(unit), k]`` rows of this rank's units (``dtype`` int64, or float64 - the bit
        patterns travel).  Returns ``[n_intervals, k]`` in the intervals' input order, on every rank."""
        as_float = np.dtype(dtype) == np.float64
        send = {u: (np.ascontiguousarray(v, dtype=np.float64).view(np.int64) if as_float
                    else np.ascontiguousarray(v, dtype=np.int64)).reshape(-1, k) for u, v in local.items()}
        n_rows = {(c, i0, i1): i1 - i0 for _, c, i0, i1 in self.units}
        got = gather_unit_rows(send, self.units, n_rows, k, group=self.group)
        out = np.zeros((self.n, k), np.int64)
        for c, rows in got.items():
            out[self.order[c]] = rows
        return out.view(np.float64) if as_float else out


def allreduce_sum(value: int, group=None, device=None) -> int:
    """Sum of one int64 over ranks (the genome-wide total of ``coverage(normalize=True)``)."""
    from . import comm
    g = comm.current(group)
    if g.world == 1:
        return int(value)
    return int(g.all_reduce_sum_i64(np.array([int(value)], np.int64))[0])


def contig_owner(weights: Dict[str, float], group=None) -> Tuple[int, int, Dict[str, int]]:
    """``(rank, world, owner)``: the contigs of ``weights`` dealt to the ranks of the initialised process group
    by LPT (everything on rank 0 without one).  Every rank computes the same map from the same inputs."""
    rank, world = rank_world(group)
    return rank, world, lpt_assign(weights, world)


def is_writer(group=None) -> bool:
    """True on the one rank that writes output files / stdout (rank 0; a plain single process)."""
    return rank_world(group)[0] == 0


def agree(error: Optional[BaseException] = None, group=None) -> None:
    """Collective check-point of a sharded command: every rank passes the exception its share of the work
    raised (or None).  If any rank failed, ALL ranks raise -- the failing rank its own exception, the others a
    ``RuntimeError`` naming it -- instead of the healthy ranks waiting forever in the next collective."""
    from . import comm
    g = comm.current(group)
    if g.world == 1:
        if error is not None:
            raise error
        return
    said = g.all_gather_object(None if error is None else f"{type(error).__name__}: {error}")
    if error is not None:
        raise error
    for r, msg in enumerate(said):
        if msg is not None:
            raise RuntimeError(f"rank {r} failed: {msg}")


def allgather_object(obj, group=None) -> list:
    """Small Python objects of every rank, in rank order (``[obj]`` without a group)."""
    from . import comm
    return comm.current(group).all_gather_object(obj)


def gather_payloads(local: Dict[int, bytes], owner: Sequence[int], group=None, device=None) -> Optional[List[bytes]]:
    """Byte payloads of numbered work items to rank 0: item ``k`` was produced by rank ``owner[k]``
    (``local[k]`` there).  Rank 0 returns the list of all payloads in item order, the other ranks ``None``.
    Transport: one size exchange, then one point-to-point message per sending rank (``ftk_comm_send`` / ``recv``:
    RCCL over xGMI; gloo in CPU tests) -- compressed output sections, a few hundred MB at most."""
    from . import comm
    g = comm.current(group)
    rank, world = g.rank, g.world
    n = len(owner)
    if world == 1:
        return [local[k] for k in range(n)]
    mine = [k for k in range(n) if owner[k] == rank]
    sizes = g.all_gather_object([len(local[k]) for k in mine])
    if rank != 0:
        blob = np.frombuffer(b"".join(bytes(local[k]) for k in mine), dtype=np.uint8).copy()
        if len(blob):
            g.send_bytes(blob, 0)
        return None
    out: List[Optional[bytes]] = [None] * n
    for k in mine:
        out[k] = bytes(local[k])
    for r in range(1, world):
        total = int(sum(sizes[r]))
        buf = g.recv_bytes(total, r) if total else np.empty(0, np.uint8)
        off = 0
        items = [k for k in range(n) if owner[k] == r]
        for k, sz in zip(items, sizes[r]):
            out[k] = buf[off:off + sz].tobytes()
            off += sz
    return out


def gather_float_rows(local: Dict[str, np.ndarray], names: Sequence[str], n_rows: Dict[str, int],
                      owner: Dict[str, int], k: int, group=None) -> Dict[str, np.ndarray]:
    """``gather_bin_vectors`` for float64 rows (bit patterns travel as int64, so every rank holds exactly the
    numbers the owner computed)."""
    as_int = {c: np.ascontiguousarray(v, dtype=np.float64).view(np.int64).reshape(-1, k) for c, v in local.items()}
    got = gather_bin_vectors(as_int, names, n_rows, {c: 1.0 for c in names}, group=group, k=k, owner=owner)
    return {c: np.ascontiguousarray(v, dtype=np.int64).view(np.float64).reshape(-1, k) for c, v in got.items()}
