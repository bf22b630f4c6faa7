#!/bin/bash
# A/B of the streaming decoder's piece size (FTK_STREAM_PIECE, compressed bytes per piece; default 48 MB): a launch of the
# inflate kernel lasts one block's chain whatever its size until the chip is full (5 120 wavefronts), so a piece of fewer
# blocks leaves wave slots idle unless enough pieces overlap.  usage: tools/piece_size_ab.sh <out dir under gpurun_out/>
OUT=gpurun_out/$1
mkdir -p $OUT
for mb in 48 96 144; do
  export FTK_STREAM_PIECE=$((mb << 20))
  FTK_DECODE_TIMING=1 python tools/bam_big_run.py > $OUT/bam_big_p$mb.json 2> $OUT/bam_big_p$mb.err
  FTK_E2E_REPS=4 python tools/e2e_genome_bench.py all 30 12 delfi > $OUT/genome_p$mb.json 2>/dev/null
done
python - "$OUT" <<'PY'
import json, sys
out = sys.argv[1]
for mb in (48, 96, 144):
    d = json.loads(open(f"{out}/bam_big_p{mb}.json").read().strip().splitlines()[-1])
    print("bam  piece", mb, "MB:", [(r["total_s"], r["decode_until_resident_s"]) for r in d["reps"]], d.get("results_ok"))
    for l in open(f"{out}/genome_p{mb}.json"):
        g = json.loads(l)
    print("text piece", mb, "MB:", [g[k]["end_to_end_s"] for k in sorted(g) if k.startswith("rep")])
PY
