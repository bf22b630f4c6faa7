#!/bin/bash
# The lane-parallel symbol loop of the device inflate (FTK_INFLATE_LANES=1) against zlib on the whole inflate suite, then
# its kernel durations beside the windowed loop's (text: contig 21 = fewer blocks than the chip holds, contig 1 =
# chip-filling; BAM records).  usage (repo root on the GPU box): bash tools/lanes_check.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
FTK_INFLATE_LANES=1 timeout 1500 python -m pytest $R/tests/test_gpu_inflate.py -x -q -m gpu 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
for lanes in 0 1; do
  export FTK_INFLATE_LANES=$lanes
  for t in "inflate_bench 21" "inflate_bench 1" "bam_inflate_probe"; do
    set -- $t
    if [ "$1" = "bam_inflate_probe" ]; then export FTK_INFLATE_VECTOR_MATCHES=1; else export FTK_INFLATE_VECTOR_MATCHES=0; fi
    rm -rf $R/gpurun_out/iv
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/iv -- python3 $R/tools/$1.py $2 > $R/gpurun_out/iv.log 2>&1
    python3 - "$R/gpurun_out/iv" "lanes=$lanes $t" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "bgzf_inflate" in r["Name"]:
        print(sys.argv[2], r["Calls"], "calls, avg", round(float(r["AverageNs"]) / 1e6, 3), "min", round(float(r["MinNs"]) / 1e6, 3), "max", round(float(r["MaxNs"]) / 1e6, 3), "ms")
PY
  done
done
