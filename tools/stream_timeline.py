#!/usr/bin/env python3
"""Occupancy of the GPU over ONE pass of a decoder stream, from a `rocprofv3 --kernel-trace --output-format csv`
directory of `rocprofv3 --kernel-trace -d <dir> -- python3 tools/first_pass_probe.py text|bam N` (the passes are told apart by the pauses between them; the last
one is analysed): how long any kernel is running, how many inflate launches run side by side and for how long, the
start and duration of every inflate launch and every row-parser launch, kernel time by name.
usage: tools/stream_timeline.py <trace dir>"""
import csv
import glob
import os
import re
import sys

f = max(glob.glob(os.path.join(sys.argv[1], "*", "*_kernel_trace.csv")), key=os.path.getmtime)


def short(k):
    m = re.search(r"(\w+_kernel|copyBuffer|fillBuffer\w*)", k)
    return m.group(1) if m else k[:30]


ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(f)))
passes = [[ev[0]]]
for a in ev[1:]:
    if a[0] - max(e[1] for e in passes[-1]) > 6e6:
        passes.append([a])
    else:
        passes[-1].append(a)
P = passes[-1]
t0, t1 = P[0][0], max(e[1] for e in P)
print(f"passes (kernels, ms): {[(len(p), round((max(e[1] for e in p) - p[0][0]) / 1e6, 1)) for p in passes]}")
busy, (cs, ce), gaps = 0, P[0][:2], []
for s, e, _ in P[1:]:
    if s > ce:
        busy += ce - cs
        gaps.append((ce - t0, s - ce))
        cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"last pass: {(t1 - t0) / 1e6:.1f} ms, some kernel running for {busy / 1e6:.1f} ms ({100 * busy / (t1 - t0):.0f} %)")
print("largest gaps (at ms, ms):", sorted([(round(a / 1e6, 1), round(b / 1e6, 2)) for a, b in gaps], key=lambda x: -x[1])[:6])
inf = [(s, e) for s, e, n in P if "inflate" in n]
pts = sorted([(s, 1) for s, _ in inf] + [(e, -1) for _, e in inf])
c, last, hist = 0, t0, {}
for t, d in pts:
    hist[c] = hist.get(c, 0) + (t - last)
    last, c = t, c + d
hist[0] = hist.get(0, 0) + (t1 - last)
print(f"inflate launches: {len(inf)}, durations add up to {sum(e - s for s, e in inf) / 1e6:.1f} ms; ms with k of them running:",
      {k: round(v / 1e6, 1) for k, v in sorted(hist.items())})
for what in ("inflate", "lines_rows", "bam_emit"):
    rows = [(s, e) for s, e, n in P if what in n]
    if rows:
        print(f"{what} (start ms : duration ms):", " ".join(f"{(s - t0) / 1e6:.1f}:{(e - s) / 1e6:.2f}" for s, e in rows))
by = {}
for s, e, n in P:
    by[n] = by.get(n, 0) + (e - s)
print("kernel time by name (ms):", {k: round(v / 1e6, 1) for k, v in sorted(by.items(), key=lambda x: -x[1])[:12]})
