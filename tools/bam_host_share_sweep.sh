#!/bin/bash
# The chr1-sized 60x BAM pass with the host threads inflating every 16th piece (default), every 32nd, none.
OUT=gpurun_out/$1
: > $OUT
for hs in 16 0 32 16 0 32; do
  FTK_BAM_HOST_SHARE=$hs FTK_BIG_REPS=4 python tools/bam_big_run.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('host_share', $hs, [r['total_s'] for r in d['reps']], d['reps'][-1]['decoder_producer_stage_ms'], d.get('results_ok'))" >> $OUT
done
cat $OUT
