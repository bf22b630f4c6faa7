#!/bin/bash
# Kernel durations of the lane-parallel inflate loop in alternative builds of the library (finaletoolkit_amd/libftk_var_*.so,
# see DESIGN 3.5c) beside the shipped one: text images of contig 21 / 1 and BAM records.
# usage (repo root on the GPU box): bash tools/lanes_variants.sh [variant names...]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
export FTK_INFLATE_LANES=1
for v in hip "$@"; do
  if [ "$v" = hip ]; then unset FTK_LIB; else export FTK_LIB=$R/finaletoolkit_amd/libftk_var_$v.so; fi
  for t in "inflate_bench 21" "inflate_bench 1" "bam_inflate_probe"; do
    set -- $t
    if [ "$1" = "bam_inflate_probe" ]; then export FTK_INFLATE_VECTOR_MATCHES=1; else export FTK_INFLATE_VECTOR_MATCHES=0; fi
    rm -rf $R/gpurun_out/iv
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/iv -- python3 $R/tools/$1.py $2 > $R/gpurun_out/iv.log 2>&1 || { echo "$v $t FAILED"; tail -3 $R/gpurun_out/iv.log; continue; }
    python3 - "$R/gpurun_out/iv" "$v $t" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "bgzf_inflate" in r["Name"]:
        print(sys.argv[2], r["Calls"], "calls, avg", round(float(r["AverageNs"]) / 1e6, 3), "min", round(float(r["MinNs"]) / 1e6, 3), "max", round(float(r["MaxNs"]) / 1e6, 3), "ms")
PY
  done
done
