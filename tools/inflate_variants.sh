#!/bin/bash
# A/B of library builds with different inflate table roots (FTK_LIB=<so>): kernel durations on the text image of contig 21
# (fewer blocks than the chip holds), of contig 1 (chip-filling) and on the BAM image.  usage: tools/inflate_variants.sh <lib suffix> ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export FTK_LIB=$R/finaletoolkit_amd/libftk_$v.so
  for t in "inflate_bench 21" "inflate_bench 1" "bam_inflate_probe"; do
    set -- $t
    # (the launch shape each kind of stream uses: BAM records with a window's matches resolved side by side)
    if [ "$1" = "bam_inflate_probe" ]; then export FTK_INFLATE_VECTOR_MATCHES=1; else export FTK_INFLATE_VECTOR_MATCHES=0; fi
    rm -rf $R/gpurun_out/iv
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/iv -- python3 $R/tools/$1.py $2 > $R/gpurun_out/iv.log 2>&1
    python3 - "$R/gpurun_out/iv" "$v $t" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "bgzf_inflate" in r["Name"]:
        print(sys.argv[2], r["Calls"], "calls, avg", round(float(r["AverageNs"]) / 1e6, 3), "max", round(float(r["MaxNs"]) / 1e6, 3), "ms")
PY
  done
done
