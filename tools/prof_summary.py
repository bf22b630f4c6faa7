#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (kernel stats / PMC counter collection) into
short text summaries for profiles/.  Usage:
    tools/prof_summary.py stats <dir> > profiles/<name>.txt
    tools/prof_summary.py pmc <dir> [<dir> ...] > profiles/<name>.txt
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def short(name):
    name = name.replace("(anonymous namespace)::", "").split("(")[0]
    for p in ("void ", "ftk::"):
        name = name.replace(p, "")
    return name[:70]


def stats(d):
    f = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    print(f"# rocprofv3 --kernel-trace --stats  ({os.path.basename(f)})")
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>10s} {'max_us':>10s} {'pct':>6s}")
    for r in rows:
        print(f"{short(r['Name']):70s} {int(r['Calls']):7d} {float(r['TotalDurationNs'])/1e6:10.3f} "
              f"{float(r['AverageNs'])/1e3:10.2f} {float(r['MinNs'])/1e3:10.2f} {float(r['MaxNs'])/1e3:10.2f} "
              f"{float(r['Percentage']):6.2f}")


def pmc(dirs):
    agg = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("# rocprofv3 --pmc (one counter per pass); per-dispatch mean, dispatch count, sum")
    print(f"{'kernel':70s} {'counter':>14s} {'n':>6s} {'mean':>16s} {'sum':>18s}")
    for k in sorted(agg):
        if k.startswith("at::") or "at::native" in k or k.startswith("rocprim") or "Cijk" in k:
            continue
        for c, v in sorted(agg[k].items()):
            print(f"{k:70s} {c:>14s} {len(v):6d} {sum(v)/len(v):16.3f} {sum(v):18.3f}")


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2])
    else:
        pmc(sys.argv[2:])
