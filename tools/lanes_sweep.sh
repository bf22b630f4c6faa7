#!/bin/bash
# Where the lane-parallel inflate loop stops paying: kernel durations (rocprofv3 --kernel-trace --stats) of both loops on
# text images of growing block counts.  usage (repo root on the GPU box): bash tools/lanes_sweep.sh [contigs...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export FTK_INFLATE_VECTOR_MATCHES=0
for c in ${@:-22 19 18 13}; do
  for lanes in 0 1; do
    export FTK_INFLATE_LANES=$lanes
    rm -rf $R/gpurun_out/iv
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/iv -- python3 $R/tools/inflate_bench.py $c > $R/gpurun_out/iv.log 2>&1
    python3 - "$R/gpurun_out/iv" "contig $c lanes=$lanes" "$(grep -o '[0-9]* blocks' $R/gpurun_out/iv.log | head -1)" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "bgzf_inflate" in r["Name"]:
        print(sys.argv[2], sys.argv[3], r["Calls"], "calls, avg", round(float(r["AverageNs"]) / 1e6, 3), "min", round(float(r["MinNs"]) / 1e6, 3), "max", round(float(r["MaxNs"]) / 1e6, 3), "ms")
PY
  done
done
