#!/bin/bash
# The text decode pass under load-time switches, one file, one box, a fresh process per setting.
# usage: tools/piece_sweep.sh <out file under gpurun_out/>
OUT=gpurun_out/$1
F=/tmp/ftk_sweep_genome.frag.gz
python tools/decode_pass.py write $F 1 > $OUT 2>/dev/null
for rep in 1 2; do
  for mb in 48 40 56 64 72 80; do
    FTK_STREAM_PIECE=$((mb << 20)) python tools/decode_pass.py run $F 5 >> $OUT 2>/dev/null
  done
done
rm -f $F $F.tbi
