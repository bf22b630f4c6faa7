#!/usr/bin/env python3
"""BASELINE config 5's shape end to end: ONE whole-genome 30x frag.gz on disk -> streaming decode -> H2D ->
coverage + length histogram + DELFI per 100 kb bin and WPS for every base -> results in host memory, contig by
contig (decode of contig k+1 runs while contig k is on the GPU; every contig's results are dropped after they
have arrived and been checked, as a writer would after writing them; the copy-back of contig k overlaps the work
on contig k+1: Engine.wps_async).
usage: tools/e2e_genome_bench.py [contigs=all] [depth=30] [workers=12]
The file is written by `workers` processes (row ranges of a contig -> BGZF pieces, concatenated in order)."""
import io
import json
import multiprocessing as mp
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ROWS_PER_TASK = 4_000_000


def write_piece(task):
    """One row range of one contig as a BGZF piece; returns (path, rows, rows with mapq >= 30 if first piece)."""
    import pandas as pd
    from finaletoolkit_amd import bgzf, synth
    contig, depth, lo, hi, path = task
    names = list(synth.B37_SIZES)
    s, e, q, st = synth.synth_contig(synth.B37_SIZES[contig], depth, synth.SEED_BASE + names.index(contig))
    truth = int((q >= 30).sum()) if lo == 0 else 0
    n = len(s)
    hi = min(hi, n)
    buf = io.StringIO()
    pd.DataFrame({"c": contig, "s": s[lo:hi], "e": e[lo:hi], "q": q[lo:hi], "t": np.where(st[lo:hi] == 1, "+", "-")}).to_csv(
        buf, sep="\t", header=False, index=False)
    data = buf.getvalue().encode()
    bgzf.write_bgzf(path, data, level=1)
    return path, hi - lo, truth, len(data), n


def main():
    from finaletoolkit_amd import synth
    contigs = (sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "all" else ",".join(synth.B37_SIZES)).split(",")
    depth = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 12
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "genome.frag.gz")
    t0 = time.time()
    tasks = []
    for c in contigs:
        n = synth.n_fragments(synth.B37_SIZES[c], depth)
        for k, lo in enumerate(range(0, n, ROWS_PER_TASK)):
            tasks.append((c, depth, lo, lo + ROWS_PER_TASK, os.path.join(tmp, f"piece.{c}.{k:03d}")))
    with mp.get_context("spawn").Pool(workers) as pool:
        done = pool.map(write_piece, tasks, chunksize=1)
    truth, rows, text_bytes = {}, 0, 0
    with open(path, "wb") as out:
        for (c, *_), (piece, n_rows, t, nbytes, _) in zip(tasks, done):
            with open(piece, "rb") as fh:
                while True:
                    b = fh.read(64 << 20)
                    if not b:
                        break
                    out.write(b)
            os.unlink(piece)
            truth[c] = truth.get(c, 0) + t
            rows += n_rows
            text_bytes += nbytes
    open(path + ".tbi", "wb").close()
    res = {"contigs": len(contigs), "fragments": rows, "text_GB": round(text_bytes / 1e9, 2),
           "file_GB": round(os.path.getsize(path) / 1e9, 2), "write_s": round(time.time() - t0, 1)}
    print(json.dumps(res), flush=True)

    from finaletoolkit_amd import source
    from finaletoolkit_amd.source import usable_cores
    threads = usable_cores()
    eng = source.get_engine()
    n_win_total = sum(-(-synth.B37_SIZES[c] // 100_000) for c in contigs)
    bases_total = sum(synth.B37_SIZES[c] for c in contigs)
    for rep in range(2):
        t1 = time.perf_counter()
        marks = []
        wait_s = compute_s = 0.0
        tb = t1
        # one result in flight behind the loop: contig k's scores travel to the host (copy stream) while contig
        # k+1 is awaited, loaded and scored; every result is checked and dropped once it has arrived
        def finish(p):
            c, size, r, w, tok = p
            eng.result_wait(tok)
            assert int(r["coverage"].sum()) == truth[c] and len(w) == size and int(w[size // 2]) == int(w[size // 2]), c
        pending = None
        for src, c in source.stream_source(path, threads):
            ta = time.perf_counter()
            wait_s += ta - tb
            size = synth.B37_SIZES[c]
            ws, we = synth.tiling_windows(size, 100_000)
            r = eng.window_features(src.key(c), ws, we, 30, hist=(0, 1001), delfi=dict(quality_threshold=30))
            w, tok = eng.wps_async(src.key(c), 0, size, size)
            if pending is not None:
                finish(pending)
            pending = (c, size, r, w, tok)
            del w, r
            tb = time.perf_counter()
            compute_s += tb - ta
            marks.append((c, round(ta - t1, 3), round(tb - t1, 3)))
        if pending is not None:
            finish(pending)
            pending = None
        dt = time.perf_counter() - t1
        res[f"rep{rep}"] = {"end_to_end_s": round(dt, 3), "windows_per_s": round(n_win_total / dt, 1),
                            "fragments_per_s_M": round(rows / dt / 1e6, 1), "text_GB_per_s": round(text_bytes / dt / 1e9, 2),
                            "wps_bases_to_host_GB_per_s": round(8 * bases_total / dt / 1e9, 2), "threads": threads,
                            "waiting_for_contigs_s": round(wait_s, 3), "kernels_and_copy_back_s": round(compute_s, 3),
                            "contig_resident_at_s / results_on_host_at_s": marks}
        source.close_all()
        eng = source.get_engine()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
