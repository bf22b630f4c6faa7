"""
Per-base features over many intervals -> one output file, on one GPU or on one rank per GPU.

``multi_wps`` and ``multi_cleavage_profile`` (reference: ``frag/_multi_wps.py:196-198,300-341``,
``frag/_cleavage_profile.py:372-395,455-500``) fan their intervals out over ``Pool(workers)`` and the parent
writes the file in interval order.  Here the intervals are grouped into RUNS (consecutive intervals on one
contig), the runs cut into UNITS of at most ``UNIT_BASES`` bases (one kernel launch, one compressed payload each), and
the units dealt to the ranks of the process group as equal-cost consecutive groups (the partition of
``sharding.split_counts`` with bases for weights: whole contigs plus at most two partial ones per rank, a partial
share scored from a REGION of the contig read through the index; no data-path exchange: a base's score depends only
on its own contig's fragments).  Every rank scores, formats and compresses ITS units -- bigWig data sections or gzip
members, so the host-side work is spread as well -- and rank 0 receives the compressed pieces
(``sharding.gather_payloads``) and lays them into the file in unit order.  A single process takes the same path with
itself as the only rank, streaming unit by unit; the units do not depend on the number of ranks, which is what makes
the multi-rank file byte-identical to the single-process one.
"""
from __future__ import annotations

import os
import pickle
from typing import Callable, List, Sequence, Tuple

import numpy as np

from .. import sharding
from .._stages import Stages

LAST_STAGE_S: dict = {}  # the last write_per_base_runs call's wall time by stage (seconds)

Run = Tuple[str, int, int]  # contig, first interval, one past the last interval


def group_runs(contigs: Sequence[str]) -> List[Run]:
    runs, i, n = [], 0, len(contigs)
    while i < n:
        j = i
        while j < n and contigs[j] == contigs[i]:
            j += 1
        runs.append((contigs[i], i, j))
        i = j
    return runs


# per-base scores one unit holds at most (one launch, one compressed payload): small enough that a few thousand sites
# deal evenly over eight ranks, large enough that a launch (~20 us) and a payload's framing do not show.
# ``FTK_UNIT_BASES`` overrides it (tests on contigs smaller than one unit) - for EVERY rank count alike, or the files of
# different rank counts would differ in their compressed pieces.
UNIT_BASES = int(os.environ.get("FTK_UNIT_BASES") or (1 << 20))


def split_into_units(runs: Sequence[Run], starts, stops) -> List[Run]:
    """Runs cut into UNITS of consecutive intervals of at most ``UNIT_BASES`` bases (an interval longer than that is a
    unit of its own).  A function of the intervals alone - never of the number of ranks - so the compressed pieces of
    the output file, and with them its bytes, are the same however many ranks produce them."""
    units = []
    for c, i, j in runs:
        a, acc = i, 0
        for k in range(i, j):
            n = max(int(stops[k]) - int(starts[k]), 0)
            if k > a and acc + n > UNIT_BASES:
                units.append((c, a, k))
                a, acc = k, 0
            acc += n
        units.append((c, a, j))
    return units


def deal_units(units: Sequence[Run], starts, stops, world: int) -> List[int]:
    """Rank of every unit: the units laid end to end in file order and cut into ``world`` consecutive groups of equal
    cost - the partition of ``sharding.split_weighted``: a unit costs the bases its intervals span (gaps included: a
    region read decodes every row in between) plus, for the first unit of a run, the fixed cost of entering a contig -
    so a rank scores whole contigs plus at most two partial ones."""
    enter = sharding.shard_overhead_bases()
    cost = []
    for k, (c, i, j) in enumerate(units):
        span = (max(int(b) for b in stops[i:j]) - min(int(a) for a in starts[i:j])) if j > i else 0
        first = k == 0 or units[k - 1][0] != c or units[k - 1][2] != i  # (a run's first unit: the contig is entered here)
        cost.append(max(span, 1) + (enter if first else 0))
    total = sum(cost)
    owner, done = [], 0
    for w in cost:
        owner.append(min(world - 1, done * world // max(total, 1)))
        done += w
    return owner


def write_per_base_runs(output_file: str, kind: str, header, contigs, starts, stops,
                        compute: Callable[[str, str, list, list], tuple], src=None, pad: int = 1) -> None:
    """``kind``: ``"bw"`` (fixedStep bigWig) or ``"bedgraph.gz"`` (``contig pos pos+1 value`` rows, gzip).
    ``compute(key, contig, starts, stops) -> (values, offsets)`` scores one unit of intervals on this rank's GPU from
    the fragment table ``key``: the contig's (``src.require``) when the rank scores all of the contig's units, else a
    REGION of it spanning the rank's intervals ``pad`` bases either side (``src.require_region``)."""
    from .. import writers
    from ..bigwig import FixedStepBigWigWriter, RunOrder, fixed_step_payload, select_intervals

    clock = Stages()
    runs = group_runs(contigs)
    units = split_into_units(runs, starts, stops)
    rank, world = sharding.rank_world()
    owner = deal_units(units, starts, stops, world)
    writer = rank == 0

    # bigWig: which intervals pyBigWig would accept depends on their coordinates only -- decided up front, on
    # every rank alike, so a rank scores exactly what will be written
    keeps = None
    if kind == "bw":
        order = RunOrder(header, quiet=not writer)
        keeps = [order.keep(c, starts[i:j], [b - a for a, b in zip(starts[i:j], stops[i:j])]) for c, i, j in units]
        chrom_id = order.chrom_id

    # the table a unit is scored from: the rank's share of a contig is one span of its units (they are consecutive)
    share: dict = {}  # contig -> [n units of the contig, n of them mine, lowest start, highest stop of mine]
    for k, (c, i, j) in enumerate(units):
        e = share.setdefault(c, [0, 0, None, None])
        e[0] += 1
        if owner[k] == rank and j > i:
            e[1] += 1
            lo, hi = min(starts[i:j]), max(stops[i:j])
            e[2] = lo if e[2] is None else min(e[2], lo)
            e[3] = hi if e[3] is None else max(e[3], hi)
    keys: dict = {}

    def table(c):
        if c not in keys:
            n_all, n_mine, lo, hi = share[c]
            clock.lap("other")
            if world == 1 or n_mine == n_all or lo is None or hi <= lo or not hasattr(src, "require_region"):
                keys[c] = src.require(c)
            else:
                keys[c] = src.require_region(c, max(0, int(lo) - pad), int(hi) + pad)
            clock.lap("decode_wait")
        return keys[c]

    def payload(k: int):
        clock.lap("file_write" if k else "plan")  # (between two payloads the consumer lays the previous one into the file)
        c, i, j = units[k]
        if kind == "bw":
            if not keeps[k]:
                return b""
            st = [starts[i + q] for q in keeps[k]]
            sp = [stops[i + q] for q in keeps[k]]
            key = table(c)
            values, offsets = compute(key, c, st, sp)
            clock.lap("score_and_copy_back")
            blob, table_, stats = fixed_step_payload(chrom_id[c], st, values, offsets)
            clock.lap("sections_compress")
            return pickle.dumps((chrom_id[c], table_, stats), protocol=4) + blob if world > 1 else (chrom_id[c], blob, table_, stats)
        key = table(c)
        values, offsets = compute(key, c, starts[i:j], stops[i:j])
        clock.lap("score_and_copy_back")
        parts = []
        for rows in writers.bedgraph_batches(c, starts[i:j], values, offsets):
            with rows:
                clock.lap("format_rows")
                if rows.n:
                    parts.append(rows.gzip_bytes(writers.GZIP_LEVEL))
                clock.lap("gzip")
        return b"".join(parts)

    def unpack_bw(p):
        if isinstance(p, tuple):
            return p
        if not p:
            return None
        import io
        fh = io.BytesIO(p)
        cid, table, stats = pickle.load(fh)
        return cid, p[fh.tell():], table, stats

    def lay_down(payloads):
        """rank 0: payloads (an iterable in run order) into the file"""
        if kind == "bw":
            with FixedStepBigWigWriter(output_file, header) as bw:
                for p in payloads:
                    got = unpack_bw(p)
                    if got is not None:
                        bw.add(*got)
            return
        writers.write_text(output_file, b"", writers.GZIP_LEVEL)  # gzip.open(..., "wt") of nothing: a valid empty file
        first = True
        for p in payloads:
            if p:
                with open(output_file, "wb" if first else "ab") as fh:
                    fh.write(p)
                first = False

    if world == 1:
        lay_down(payload(k) for k in range(len(units)))  # streamed: one unit in memory at a time
        clock.lap("file_write")
        clock.publish(LAST_STAGE_S)
        return
    local, err = {}, None
    try:
        for k in range(len(units)):
            if owner[k] == rank:
                local[k] = payload(k)
    except Exception as e:  # noqa: BLE001 - handed to every rank below
        err = e
    finally:
        for key in keys.values():
            if hasattr(src, "release_region"):
                src.release_region(key)  # (a no-op for a whole contig's key)
    sharding.agree(err)
    got = sharding.gather_payloads(local, owner)
    err = None
    if writer:
        try:
            lay_down(got)
        except Exception as e:  # noqa: BLE001
            err = e
    sharding.agree(err)
