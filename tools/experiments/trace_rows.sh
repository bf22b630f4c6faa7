#!/bin/bash
# rocprofv3 kernel trace of tools/kernel_rows.py: per-kernel durations next to the event-timed rows
# usage: bash tools/experiments/trace_rows.sh <out name under gpurun_out/> [next|bam] [grep pattern]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
W=${2:-next}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/kt -- python3 $GRAFT_REPO_ROOT/tools/kernel_rows.py $W 5 > $OUT/rows.json 2> $OUT/kt.err
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py stats $OUT/kt > $OUT/kernel_stats.txt 2>/dev/null || python tools/prof_summary.py $OUT/kt > $OUT/kernel_stats.txt
rm -rf $OUT/kt
grep -i "${3:-feat\|plan\|gc_count\|adjust\|cleav}" $OUT/kernel_stats.txt | cut -c1-160
