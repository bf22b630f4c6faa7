#!/usr/bin/env python3
"""Timeline of a rocprofv3 csv trace directory (kernel + memory-copy + optional hip-api traces): the last `n`
inflate launches with everything that takes more than 50 us around them.
usage: tools/trace_view.py <dir> [n=12]"""
import csv
import glob
import os
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12


def newest(pat):
    f = glob.glob(os.path.join(d, "*", pat))
    return max(f, key=os.path.getmtime) if f else None


kt = list(csv.DictReader(open(newest("*_kernel_trace.csv"))))
inf = [r for r in kt if "bgzf_inflate" in r["Kernel_Name"]]
t0 = int(inf[-n]["Start_Timestamp"]) - 2_000_000
ev = []
for r in kt:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s >= t0:
        ev.append((s, e, "K  " + r["Kernel_Name"].split("(")[0][-28:] + " s" + r["Stream_Id"]))
f = newest("*_memory_copy_trace.csv")
if f:
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s >= t0:
            ev.append((s, e, "C  " + r["Direction"][12:] + " s" + r["Stream_Id"]))
f = newest("*_hip_api_trace.csv")
if f:
    tid = inf[-1]["Thread_Id"]
    for r in csv.DictReader(open(f)):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if s >= t0 and r["Thread_Id"] == tid:
            ev.append((s, e, "   api " + r["Function"]))
ev.sort()
for s, e, name in ev:
    if e - s > 50_000:
        print(f"{(s - t0) / 1e6:9.3f} {(e - s) / 1e6:8.3f} {name}")
