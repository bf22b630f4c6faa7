#!/bin/bash
# Both symbol loops of bgzf_inflate_kernel alone, launch by launch (rocprofv3 kernel trace): tools/inflate_bench.py on contig
# 21 (1 882 blocks: fewer than the chip holds) and contig 1 (10 341 blocks: chip-filling) - libdeflate levels 1 and 6, three
# calls each, then a 40 MB zlib-6 part - and 590 MB of 60x BAM records (tools/bam_inflate_probe.py).
# usage (repo root on the GPU box): bash tools/inflate_loops.sh > profiles/rN_inflate_loops.txt
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
for lanes in 1 0; do
  export FTK_INFLATE_LANES=$lanes
  for t in "inflate_bench 21" "inflate_bench 1" "bam_inflate_probe"; do
    set -- $t
    if [ "$1" = "bam_inflate_probe" ]; then export FTK_INFLATE_VECTOR_MATCHES=1; else export FTK_INFLATE_VECTOR_MATCHES=0; fi
    rm -rf $R/gpurun_out/il
    rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/il -- python3 $R/tools/$1.py $2 > $R/gpurun_out/il.log 2>&1
    echo "== FTK_INFLATE_LANES=$lanes ($([ $lanes = 1 ] && echo 'lane-parallel loop, the default' || echo 'windowed loop')) tools/$1.py $2"
    grep -E "MB ->|blocks" $R/gpurun_out/il.log | sed -n '1p;4p;7p'
    python3 - "$R/gpurun_out/il" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_trace.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if "bgzf_inflate" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
for k, r in enumerate(rows):
    print("  launch", k, "workgroups", int(r.get("Grid_Size_X", r.get("Grid_Size", 0))) // max(int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 64))), 1), "ms", round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 3))
PY
  done
done
