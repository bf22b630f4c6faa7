#!/bin/bash
# The chr1-sized 60x BAM pass (tools/bam_big_run.py) under FTK_READ_THREADS = 4 / 8 / 16: is the pass paced by the copy
# out of the page cache?   usage: tools/read_threads_sweep.sh <out file under gpurun_out/>
OUT=gpurun_out/$1
: > $OUT
for rt in 4 8 16 4 8 16; do
  FTK_READ_THREADS=$rt FTK_BIG_REPS=4 python tools/bam_big_run.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('read_threads', $rt, [r['total_s'] for r in d['reps']], d['reps'][-1]['decoder_producer_stage_ms'], d.get('results_ok'))" >> $OUT
done
cat $OUT
