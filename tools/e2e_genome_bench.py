#!/usr/bin/env python3
"""BASELINE config 5's shape end to end: ONE whole-genome 30x frag.gz on disk -> streaming decode -> H2D ->
coverage + length histogram + DELFI per 100 kb bin and WPS for every base -> results in host memory, contig by
contig (decode of contig k+1 runs while contig k is on the GPU; every contig's results are dropped after they
have arrived and been checked, as a writer would after writing them; the copy-back of contig k overlaps the work
on contig k+1: Engine.wps_async).
usage: tools/e2e_genome_bench.py [contigs=all] [depth=30] [workers=12] [delfi]   (delfi: DELFI bins only, no per-base WPS)
The file is written by `workers` processes (row ranges of a contig -> BGZF pieces, concatenated in order)."""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from finaletoolkit_amd import synth, writers
    contigs = (sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "all" else ",".join(synth.B37_SIZES)).split(",")
    depth = float(sys.argv[2]) if len(sys.argv) > 2 else 30.0
    no_wps = len(sys.argv) > 4 and sys.argv[4] == "delfi"
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "genome.frag.gz")
    t0 = time.time()
    # fragments from the device generator (the bench's), rows formatted and BGZF blocks deflated by the library's
    # host threads, one contig after the other appended to the file
    dev = torch.device("cuda", 0)
    names = list(synth.B37_SIZES)
    truth, rows, text_bytes = {}, 0, 0
    for k, c in enumerate(contigs):
        size = synth.B37_SIZES[c]
        n = synth.n_fragments(size, depth)
        s, e, q, st = (t.cpu().numpy() for t in synth.gen_contig_device(torch, dev, size, n, synth.SEED_BASE + names.index(c)))
        truth[c] = dict(cov=int((q >= 30).sum()), keep=(s, e, q) if c in (contigs[0], contigs[-1]) else None)
        with writers.frag_rows(c, s, e, q, st) as text:
            writers.bgzf_write(path, text, int(os.environ.get("FTK_E2E_LEVEL", "1")), append=k > 0, write_eof=k == len(contigs) - 1)
            text_bytes += text.n
        rows += n
    open(path + ".tbi", "wb").close()
    res = {"contigs": len(contigs), "fragments": rows, "text_GB": round(text_bytes / 1e9, 2),
           "file_GB": round(os.path.getsize(path) / 1e9, 2), "write_s": round(time.time() - t0, 1),
           "FTK_DEVICE_INFLATE": os.environ.get("FTK_DEVICE_INFLATE", "1"), "mode": "delfi bins only" if no_wps else "all features + WPS"}
    print(json.dumps(res), flush=True)

    from finaletoolkit_amd import source
    from finaletoolkit_amd.source import usable_cores
    threads = usable_cores()
    eng = source.get_engine()
    n_win_total = sum(-(-synth.B37_SIZES[c] // 100_000) for c in contigs)
    bases_total = sum(synth.B37_SIZES[c] for c in contigs)
    for rep in range(int(os.environ.get("FTK_E2E_REPS", "2"))):
        t1 = time.perf_counter()
        marks = []
        wait_s = compute_s = 0.0
        tb = t1
        # one result in flight behind the loop: contig k's scores travel to the host (copy stream) while contig
        # k+1 is awaited, loaded and scored; every result is checked and dropped once it has arrived
        def finish(p):
            c, size, r, w, tok = p
            eng.result_wait(tok)
            assert int(r["coverage"].sum()) == truth[c]["cov"] and len(w) == size, c
            if truth[c]["keep"] is not None:  # first and last contig: a 2 Mb stretch of scores against the closed form
                fs, fe, fq = truth[c]["keep"]
                ok = (fq >= 30) & (fe - fs >= 120) & (fe - fs <= 180)
                a, b = size // 2, size // 2 + 2_000_000
                sel = ok & (fe > a - 200) & (fs < b + 200)
                d = np.zeros(b - a + 1000, np.int64)
                for pos, val in ((fs[sel] - 59, -1), (fs[sel] + 61, 2), (fe[sel] - 59, -2), (fe[sel] + 61, 1)):
                    np.add.at(d, np.clip(pos.astype(np.int64) - a + 400, 0, len(d) - 1), val)
                assert np.array_equal(np.asarray(w[a:b]), np.cumsum(d)[400:400 + (b - a)]), c
        pending = None
        for src, c in source.stream_source(path, threads):
            ta = time.perf_counter()
            wait_s += ta - tb
            size = synth.B37_SIZES[c]
            ws, we = synth.tiling_windows(size, 100_000)
            if no_wps:  # BASELINE config 4's own shape: the DELFI bins of the whole genome, no per-base output
                sh, lg, nf = eng.delfi_counts(src.key(c), ws, we, 30, None, None, synth.synth_gaps(size))
                assert bool(np.array_equal(sh + lg, nf)), c
                tb = time.perf_counter()
                compute_s += tb - ta
                marks.append((c, round(ta - t1, 3), round(tb - t1, 3)))
                continue
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
