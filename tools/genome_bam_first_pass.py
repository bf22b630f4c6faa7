#!/usr/bin/env python3
"""bench.py's whole-genome 60x BAM leg alone, every repetition's stage split kept (FTK_BENCH_REP_STAGES=1) and the
decoder's own trail on stderr (FTK_DECODE_TIMING=1): what the FIRST pass pays that the later ones do not.
usage: python tools/genome_bam_first_pass.py > gpurun_out/x.json 2> gpurun_out/x.err"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("FTK_BENCH_REP_STAGES", "1")
os.environ.setdefault("FTK_DECODE_TIMING", "1")
import torch  # noqa: E402

import bench  # noqa: E402
from finaletoolkit_amd import _lib, source  # noqa: E402

_lib.load()
dev = torch.device("cuda", 0)
threads = source.usable_cores()
h2d = bench.measure_h2d_gbs(torch, dev)
bench.LINK["d2h_GBps"] = round(bench.measure_d2h_gbs(torch, dev), 1)
bench.LINK["host_write_GBps"] = round(bench.measure_host_write_gbs(threads), 1)
leg = bench.genome_bam_leg(torch, dev, threads, h2d, bench.inflate_alone_rates(), None, reps=3)
leg.pop("checked", None)
print(json.dumps(leg))
