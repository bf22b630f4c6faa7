#!/usr/bin/env python3
"""frag.frag_length_intervals on the whole-genome 30x file, 30 970 x 100 kb intervals: its stage laps and a per-contig
timeline (when each contig became resident, how long its kernels / rows took).  usage: python3 tools/experiments/intervals_probe.py [reps]"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from finaletoolkit_amd import _lib, frag, source, synth, writers  # noqa: E402
from finaletoolkit_amd.frag import _frag_length as FL  # noqa: E402
from finaletoolkit_amd.synth import gen_contig_device  # noqa: E402
import cmd_legs  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
if os.environ.get('NO_SWITCH'):  # A/B: leave the interpreter's switch interval alone
    sys.setswitchinterval = lambda x: None
_lib.load()
dev = torch.device("cuda", 0)
sizes = dict(synth.B37_SIZES)
tmp = tempfile.mkdtemp(prefix="ftk_ivp_")
try:
    path = os.path.join(tmp, "genome.frag.gz")
    names = list(sizes)
    for k, c in enumerate(names):
        s, e, q, st = (t.cpu().numpy() for t in gen_contig_device(torch, dev, sizes[c], synth.n_fragments(sizes[c], 30.0), synth.SEED_BASE + k))
        with writers.frag_rows(c, s, e, q, st) as text:
            writers.bgzf_write(path, text, 1, append=k > 0, write_eof=k == len(names) - 1)
    open(path + ".tbi", "wb").close()
    cs, bed, sites, n_win, n_site_rows = cmd_legs.write_side_files(tmp, sizes, 20_000)
    marks = []
    t_start = [0.0]
    real_rows, real_stats = FL._result_rows, None

    def rows(*a, **k):
        t0 = time.perf_counter()
        if not os.environ.get("NO_ROWS"):
            real_rows(*a, **k)
        marks.append(("rows", t0 - t_start[0], time.perf_counter() - t0))
    FL._result_rows = rows
    eng = FL.get_engine()
    real_stats = eng.fraglen_stats

    def stats(*a, **k):
        t0 = time.perf_counter()
        r = real_stats(*a, **k)
        marks.append(("stats", t0 - t_start[0], time.perf_counter() - t0))
        return r
    eng.fraglen_stats = stats
    out = os.path.join(tmp, "o.bed")
    for r in range(reps):
        marks.clear()
        source.close_all()
        t_start[0] = time.perf_counter()
        frag.frag_length_intervals(path, bed, None if os.environ.get('NO_ROWS') else out, quality_threshold=30)
        tot = time.perf_counter() - t_start[0]
        print("rep", r, "total %.1f ms" % (tot * 1e3), {k: round(v * 1e3, 1) for k, v in FL.LAST_STAGE_S.items()})
    print("last repetition, per call: kind, start ms, duration ms")
    print(" ".join("%s@%.1f+%.1f" % (k, a * 1e3, d * 1e3) for k, a, d in marks))
finally:
    shutil.rmtree(tmp, ignore_errors=True)
