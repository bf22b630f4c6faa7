#!/bin/bash
# Average duration of the inflate / CRC kernels on the text (tools/inflate_bench.py) and BAM (tools/bam_inflate_probe.py)
# images, from rocprofv3's kernel trace.  usage (on the GPU box): bash tools/inflate_kernel_times.sh [label]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for t in inflate_bench bam_inflate_probe; do
  rm -rf $R/gpurun_out/ikt_$t
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ikt_$t -- python3 $R/tools/$t.py > $R/gpurun_out/ikt_$t.log 2>&1
  python3 - "$R/gpurun_out/ikt_$t" "$1 $t" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "bgzf" in r["Name"]:
        print(sys.argv[2], r["Name"].split("::")[-1].split("(")[0], r["Calls"], "calls, avg", round(float(r["AverageNs"]) / 1e6, 3), "ms")
PY
done
