#!/usr/bin/env python3
"""One file, decoded again and again: `write <path> [level] [scale]` writes the bench's whole-genome 30x frag.gz once;
`run <path> [reps]` streams it through the product path (source.stream_source + DELFI bins per contig, the
`genome_delfi_bins` leg of bench.py) and prints every repetition's wall time and the producer's stage split - in a fresh
process per setting, so that load-time switches (FTK_STREAM_PIECE, FTK_TEXT_LAG, FTK_HW_QUEUES ...) can be compared on
one file and one box (tools/piece_sweep.sh)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from finaletoolkit_amd import synth  # noqa: E402


def write(path, level=1, scale=1.0):
    import torch
    from finaletoolkit_amd import writers
    dev = torch.device("cuda", 0)
    names = list(synth.B37_SIZES)
    for k, c in enumerate(names):
        size = max(int(synth.B37_SIZES[c] * scale), 200_000)
        s, e, q, st = (t.cpu().numpy() for t in synth.gen_contig_device(torch, dev, size, synth.n_fragments(size, 30.0), synth.SEED_BASE + k))
        with writers.frag_rows(c, s, e, q, st) as text:
            writers.bgzf_write(path, text, level, append=k > 0, write_eof=k == len(names) - 1)
    open(path + ".tbi", "wb").close()
    print(json.dumps(dict(file_GB=round(os.path.getsize(path) / 1e9, 3), level=level, scale=scale)))


def run(path, reps=5):
    from finaletoolkit_amd import _lib, source
    _lib.load()
    threads = source.usable_cores()
    out = []
    for _ in range(reps):
        source.close_all()
        eng = source.get_engine()
        t0 = time.perf_counter()
        src, n = None, 0
        for src, c in source.stream_source(path, threads):
            size = eng.info(src.key(c))[2] + 1
            ws, we = synth.tiling_windows(size, 100_000)
            sh, lg, nf = eng.delfi_counts(src.key(c), ws, we, 30, None, None, None)
            n += int(nf.sum())
        out.append((round(time.perf_counter() - t0, 4), src.decode_stage_ms))
    ts = [o[0] for o in out]
    print(json.dumps(dict(env={k: v for k, v in os.environ.items() if k.startswith("FTK_")}, best_s=min(ts), median_s=float(np.median(ts)),
                          all_s=ts, stages_best=out[int(np.argmin(ts))][1], delfi_fragments=n)))


if __name__ == "__main__":
    if sys.argv[1] == "write":
        write(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1, float(sys.argv[4]) if len(sys.argv) > 4 else 1.0)
    else:
        run(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 5)
