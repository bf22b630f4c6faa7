#!/usr/bin/env python3
"""Where frag.wps(chr22, output_file=.wig) spends its time (GPU box): kernel + copy back, the 80-byte record array the
reference's signature returns, the formatter, the file write.  usage: tools/wps_write_probe.py"""
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from finaletoolkit_amd import bgzf, source, synth, writers  # noqa: E402
from finaletoolkit_amd.frag import _wps as W  # noqa: E402

size = synth.B37_SIZES["22"]
tmp = tempfile.mkdtemp(prefix="ftk_wp_")
s, e, q, st = synth.synth_contig(size, 30.0, synth.SEED_BASE + 21)
path = os.path.join(tmp, "chr22.frag.gz")
bgzf.write_frag_gz(path, [("22", s, e, q, st)], level=1, with_index=True)
src = source.open_source(path)
eng = source.get_engine()
key = src.require("22")
out = {}
for rep in range(3):
    t = [time.perf_counter()]
    vals = eng.wps(key, 0, size, size)
    t.append(time.perf_counter())
    rec = W._scores_array("22", 0, vals)
    t.append(time.perf_counter())
    body = writers.wig_body(rec["wps"])
    t.append(time.perf_counter())
    writers.write_text(os.path.join(tmp, "o.wig"), b"fixedStep\n", 0)
    body.write(os.path.join(tmp, "o.wig"), 0, append=True)
    t.append(time.perf_counter())
    body.free()
    vals2 = np.ascontiguousarray(rec["wps"])
    t.append(time.perf_counter())
    out[f"rep{rep}"] = dict(zip(("kernel_and_copy_back", "record_array_80B", "format_wig", "file_write", "strided_gather_of_wps_field"),
                                (round(b - a, 4) for a, b in zip(t, t[1:]))))
    del vals, rec, body, vals2
print(json.dumps(out, indent=1))
