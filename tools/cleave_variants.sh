#!/bin/bash
# cleavage_kernel in alternative builds of the library (finaletoolkit_amd/libftk_cv_*.so: ftk_kernels.hip compiled with
# the FTK_CLEAVE_* switches of tools/experiments/kernel_experiment_switches.patch, built by tools/experiments/build_variant.sh) beside the shipped one: tools/kernel_rows.py's cleavage row (HIP events,
# chr2-sized 30x contig).  usage (repo root on the GPU box): bash tools/cleave_variants.sh [variant names...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for v in hip "$@"; do
  if [ "$v" = hip ]; then unset FTK_LIB; else export FTK_LIB=$R/finaletoolkit_amd/libftk_cv_$v.so; fi
  python3 - <<PY
import sys, json
sys.path.insert(0, "$R")
import torch
from finaletoolkit_amd.engine import Engine
from tools import kernel_rows as K
eng = Engine(0)
st = torch.cuda.Stream(); torch.cuda.set_stream(st); eng.set_stream(st.cuda_stream)
# (only the cleavage row: the other rows are skipped by cutting the function short)
import numpy as np
from finaletoolkit_amd import synth
dev = torch.device("cuda", 0)
size = synth.B37_SIZES["2"]; n = synth.n_fragments(size, 30.0)
s, e, q, stt = synth.gen_contig_device(torch, dev, size, n, 1)
torch.cuda.synchronize()
eng.load_contig_device("c", s, e, q, stt, n)
cl = torch.empty(size, dtype=torch.float64, device=dev)
w = torch.empty(size, dtype=torch.int64, device=dev)
fb = torch.empty(160_000_000, dtype=torch.int32, device=dev)
def run(f, reps=7):
    f(); eng.sync(); ts = []
    for _ in range(reps):
        fb.sum()
        eng.event_record(10); f(); eng.event_record(11); ts.append(eng.event_elapsed_ms(10, 11))
    return float(np.median(ts)), float(np.min(ts))
c = run(lambda: eng.cleavage("c", 0, size, None, None, 20, out=cl))
p = run(lambda: eng.wps("c", 0, size, size, 120, 120, 180, 30, out=w))
print("$v: cleavage median %.4f ms best %.4f ms (%.3f of 8 TB/s); wps median %.4f ms" % (c[0], c[1], (10 * n + 8 * size) / (c[0] * 1e-3) / 8e12, p[0]))
PY
done
