"""
Per-base features over many intervals -> one output file, on one GPU or on one rank per GPU.

``multi_wps`` and ``multi_cleavage_profile`` (reference: ``frag/_multi_wps.py:196-198,300-341``,
``frag/_cleavage_profile.py:372-395,455-500``) fan their intervals out over ``Pool(workers)`` and the parent
writes the file in interval order.  Here the intervals are grouped into RUNS (consecutive intervals on one
contig = one kernel launch), the contigs are dealt to the ranks of the process group (LPT on their bases; no
data-path exchange: a base's score depends only on its own contig's fragments), every rank scores, formats and
compresses the runs of ITS contigs -- bigWig data sections or gzip members, so the host-side work is spread
as well -- and rank 0 receives the compressed pieces (``sharding.gather_payloads``) and lays them into the
file in run order.  A single process takes the same path with itself as the only rank, streaming run by run,
which is what makes the multi-rank file byte-identical to the single-process one.
"""
from __future__ import annotations

import pickle
from typing import Callable, List, Sequence, Tuple

import numpy as np

from .. import sharding

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


def write_per_base_runs(output_file: str, kind: str, header, contigs, starts, stops,
                        compute: Callable[[str, list, list], tuple]) -> None:
    """``kind``: ``"bw"`` (fixedStep bigWig) or ``"bedgraph.gz"`` (``contig pos pos+1 value`` rows, gzip).
    ``compute(contig, starts, stops) -> (values, offsets)`` scores one run on this rank's GPU."""
    from .. import writers
    from ..bigwig import FixedStepBigWigWriter, RunOrder, fixed_step_payload, select_intervals

    runs = group_runs(contigs)
    weights: dict = {}
    for c, i, j in runs:
        weights[c] = weights.get(c, 0.0) + float(sum(stops[i:j]) - sum(starts[i:j])) + 1.0
    rank, world, owner = sharding.contig_owner(weights)
    writer = rank == 0

    # bigWig: which intervals pyBigWig would accept depends on their coordinates only -- decided up front, on
    # every rank alike, so a rank scores exactly what will be written
    keeps = None
    if kind == "bw":
        order = RunOrder(header, quiet=not writer)
        keeps = [order.keep(c, starts[i:j], [b - a for a, b in zip(starts[i:j], stops[i:j])]) for c, i, j in runs]
        chrom_id = order.chrom_id

    def payload(k: int) -> bytes:
        c, i, j = runs[k]
        if kind == "bw":
            if not keeps[k]:
                return b""
            st = [starts[i + q] for q in keeps[k]]
            sp = [stops[i + q] for q in keeps[k]]
            values, offsets = compute(c, st, sp)
            blob, table, stats = fixed_step_payload(chrom_id[c], st, values, offsets)
            return pickle.dumps((chrom_id[c], table, stats), protocol=4) + blob if world > 1 else (chrom_id[c], blob, table, stats)
        values, offsets = compute(c, starts[i:j], stops[i:j])
        parts = []
        for rows in writers.bedgraph_batches(c, starts[i:j], values, offsets):
            with rows:
                if rows.n:
                    parts.append(rows.gzip_bytes(writers.GZIP_LEVEL))
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
        lay_down(payload(k) for k in range(len(runs)))  # streamed: one run in memory at a time
        return
    local, err = {}, None
    try:
        for k, (c, _, _) in enumerate(runs):
            if owner[c] == rank:
                local[k] = payload(k)
    except Exception as e:  # noqa: BLE001 - handed to every rank below
        err = e
    sharding.agree(err)
    got = sharding.gather_payloads(local, [owner[c] for c, _, _ in runs])
    err = None
    if writer:
        try:
            lay_down(got)
        except Exception as e:  # noqa: BLE001
            err = e
    sharding.agree(err)
