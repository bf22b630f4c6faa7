#!/usr/bin/env python3
"""First-call latency of a one-interval API call on a contig that is not resident: the interval's rows read through the
index (FragSource.require_interval) against decoding the whole contig.  usage: tools/interval_call_probe.py [contig=1 | bam]"""
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from finaletoolkit_amd import bgzf, frag, source, synth  # noqa: E402

contig = sys.argv[1] if len(sys.argv) > 1 else "1"
if contig == "bam":  # a 60x paired-end BAM slice of 24 Mb instead (tools/bam_e2e_bench.py's file)
    contig, size = "mid", 24_000_000
    p = os.path.join(tempfile.mkdtemp(), "slice.bam")
    exp = synth.write_paired_bam(p, contig, size, 60.0, 31)
    res = {"contig": contig, "fragments": int(exp["n"]), "file_MB": round(os.path.getsize(p) / 1e6, 1), "calls": []}
else:
    size = synth.B37_SIZES[contig]
    s, e, q, st = synth.synth_contig(size, 30.0, 5)
    p = os.path.join(tempfile.mkdtemp(), "one.frag.gz")
    bgzf.write_frag_gz(p, [(contig, s, e, q, st)], level=1, with_index=True)
    res = {"contig": contig, "fragments": int(len(s)), "file_MB": round(os.path.getsize(p) / 1e6, 1), "calls": []}
a = size // 3
for label, force_whole in (("region", False), ("whole contig", True), ("region", False), ("whole contig", True)):
    source.close_all()
    src = source.open_source(p)
    if force_whole:
        src._interval_hits[contig] = 10
    t0 = time.perf_counter()
    w = frag.wps(p, contig, a, a + 20_000, size)
    dt = time.perf_counter() - t0
    res["calls"].append({"how": label, "first_call_ms": round(dt * 1e3, 2), "checksum": int(w["wps"].sum())})
assert len({c["checksum"] for c in res["calls"]}) == 1
print(json.dumps(res))
