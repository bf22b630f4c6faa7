#!/bin/bash
# kernel + memory-copy timeline of tools/kernel_rows.py: what runs, in order, with the gaps, around one kernel
# usage: bash tools/experiments/timeline_rows.sh <out name under gpurun_out/> <next|bam> <kernel name pattern> [rows before/after] [which hit, default -1 = last]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/tl -- python3 $GRAFT_REPO_ROOT/tools/kernel_rows.py $2 2 > /dev/null 2> $OUT/tl.err
cd $GRAFT_REPO_ROOT
python3 - "$OUT" "$3" "${4:-6}" "${5:--1}" <<'PY'
import csv, glob, sys
out, pat, ctx, which = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
ev = []
for f in glob.glob(out + "/tl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
for f in glob.glob(out + "/tl/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "COPY " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
ev.sort()
hits = [i for i, e in enumerate(ev) if pat in e[2]]
i = hits[which]
lines = []
for j in range(max(0, i - ctx), min(len(ev), i + ctx + 1)):
    s, e, n = ev[j]
    gap = (s - ev[j - 1][1]) / 1e3 if j else 0.0
    lines.append("%9.1f us after the previous ended | %8.1f us | %s" % (gap, (e - s) / 1e3, n))
open(out + "/timeline.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/tl
