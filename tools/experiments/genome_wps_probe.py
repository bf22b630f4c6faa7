#!/usr/bin/env python3
"""bench.py's `genome_all_features_wps` leg alone (whole-genome 30x frag.gz -> every feature + WPS of every base on the
host, contig by contig), with the library's own account of each copy-back (FTK_WPS_TIMING).
usage: FTK_WPS_TIMING=1 python3 tools/experiments/genome_wps_probe.py [reps]"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from finaletoolkit_amd import _lib, source, synth, writers  # noqa: E402
from finaletoolkit_amd.synth import gen_contig_device  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
_lib.load()
dev = torch.device("cuda", 0)
sizes = dict(synth.B37_SIZES)
threads = source.usable_cores()
tmp = tempfile.mkdtemp(prefix="ftk_gwp_")
try:
    path = os.path.join(tmp, "genome.frag.gz")
    names = list(sizes)
    for k, c in enumerate(names):
        s, e, q, st = (t.cpu().numpy() for t in gen_contig_device(torch, dev, sizes[c], synth.n_fragments(sizes[c], 30.0), synth.SEED_BASE + k))
        with writers.frag_rows(c, s, e, q, st) as text:
            writers.bgzf_write(path, text, 1, append=k > 0, write_eof=k == len(names) - 1)
    open(path + ".tbi", "wb").close()
    for r in range(reps):
        source.close_all()
        eng = source.get_engine()
        sys.stderr.write(f"--- repetition {r}\n")
        t0 = time.perf_counter()
        t_wait = t_work = 0.0
        tb = t0
        marks = []
        for src, c in source.stream_source(path, threads):
            ta = time.perf_counter()
            t_wait += ta - tb
            ws, we = synth.tiling_windows(sizes[c], 100_000)
            res, w = eng.all_features_wps(src.key(c), ws, we, sizes[c], 30, hist_bins=(0, 1001), delfi_q=30, window_size=120,
                                          wps_min_length=120, wps_max_length=180, wps_quality=30)
            del w, res
            tb = time.perf_counter()
            t_work += tb - ta
            marks.append("%s:%.0f+%.1f" % (c, (ta - t0) * 1e3, (tb - ta) * 1e3))
        print("rep %d total %.1f ms  waiting %.1f  features+wps+copy-back %.1f" % (r, (tb - t0) * 1e3, t_wait * 1e3, t_work * 1e3))
        print("  contig:start ms+duration ms  " + " ".join(marks))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
