#!/bin/bash
# bgzf_inflate_kernel alone per DEFLATE level: every launch of tools/inflate_bench.py 1 (chr1 at 30x, 650 MB of fragment
# rows, chip-filling; libdeflate levels 1 and 6, three calls each, then a 40 MB zlib-6 part) with its duration, from
# rocprofv3's kernel trace.  usage: tools/inflate_levels.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/il
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/il -- python3 $R/tools/inflate_bench.py 1 > $R/gpurun_out/il.log 2>&1
python3 - "$R/gpurun_out/il" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "bgzf_inflate" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for k, r in enumerate(rows):
    print(k, "grid", r.get("Grid_Size_X", r.get("Grid_Size", "?")), "ms", round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3))
PY
