"""ONE streamed pass over a large file in a fresh process - what a command-line call pays - with the time of every
contig's arrival, for runs under `rocprofv3 --hip-runtime-trace --stats` (which HIP calls the first pass spends its time
in: page-locking, device allocations, stream creation, code-object loads).
usage: python3 tools/first_pass_probe.py bam|text [passes=1]
under the profiler the interpreter itself stands behind `--` (an `env` / shebang hop would be an exec after the profiler's
library has initialised the GPU): rocprofv3 --kernel-trace -d <dir> -- python3 tools/first_pass_probe.py text 4"""
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from finaletoolkit_amd import bgzf, source, synth  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "bam"
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 1
# FTK_PROBE_DIR=<dir>: the file is written there once and kept (several runs - A/B of an environment variable - on one file)
keep = os.environ.get("FTK_PROBE_DIR")
tmp = keep or tempfile.mkdtemp(prefix="ftk_first_")
os.makedirs(tmp, exist_ok=True)
W = 100_000
if kind == "bam":
    contigs = [("small_a", 3_000_000), ("big", synth.B37_SIZES["1"]), ("small_c", 5_000_000)]
    path = os.path.join(tmp, "wg60x.bam")
    if not os.path.exists(path + ".bai"):
        synth.write_paired_bam_contigs(path, contigs, 60.0, 4242)
    sizes = dict(contigs)
else:
    names = list(synth.B37_SIZES)
    sizes = dict(synth.B37_SIZES)
    path = os.path.join(tmp, "genome.frag.gz")

    def rows():
        for k, c in enumerate(names):
            yield (c,) + synth.synth_contig(sizes[c], 30.0, synth.SEED_BASE + k)
    if not os.path.exists(path + ".tbi"):
        bgzf.write_frag_gz_contigs(path, rows(), level=1, with_index=False)
print(f"file {os.path.getsize(path) / 1e9:.2f} GB", file=sys.stderr)
for rep in range(passes):
    source.close_all()
    t0 = time.perf_counter()
    eng = source.get_engine()
    t_ctx = time.perf_counter() - t0
    marks = []
    for src, c in source.stream_source(path):
        ta = time.perf_counter()
        ws, we = synth.tiling_windows(sizes[c], W)
        key = src.key(c)
        if kind == "bam":
            eng.window_features(key, ws, we, 30, hist=(0, 1001), delfi=dict(quality_threshold=30))
            w = eng.wps(key, 0, sizes[c], sizes[c], 120, 120, 180, 30)
            del w
        else:
            eng.delfi_counts(key, ws, we, 30, None, None, synth.synth_gaps(sizes[c]))
        marks.append((c, round(ta - t0, 4), round(time.perf_counter() - t0, 4)))
    print(f"pass {rep}: context {t_ctx:.3f} s, total {time.perf_counter() - t0:.3f} s, stages {src.decode_stage_ms}", file=sys.stderr)
    print("  (contig, resident at, scored at):", marks[:4], "...", marks[-2:], file=sys.stderr)
source.close_all()
if not keep:
    for f in os.listdir(tmp):
        os.remove(os.path.join(tmp, f))
    os.rmdir(tmp)
