#!/bin/bash
# the next-row / BAM-kernel part of tools/profile_round.sh alone
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
K="python3 $GRAFT_REPO_ROOT/tools/kernel_rows.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/next_stats -- $K next 5 > $OUT/next_rows_under_profiler.json 2> $OUT/next_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/next_fetch -- $K next 2 > /dev/null 2> $OUT/next_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/next_write -- $K next 2 > /dev/null 2> $OUT/next_write.err
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py stats $OUT/next_stats > $OUT/nextrow_kernel_stats.txt
python tools/prof_summary.py pmc $OUT/next_fetch $OUT/next_write > $OUT/nextrow_pmc_hbm.txt
$K all 5 > $OUT/kernel_rows.json 2> $OUT/kernel_rows.err
rm -rf $OUT/next_stats $OUT/next_fetch $OUT/next_write
bash tools/pmc_kernel_rows.sh $1_pmc next > /dev/null 2>&1
grep -i "feat_large\|adjust\|gc_count\|cleav" $OUT/nextrow_kernel_stats.txt | cut -c1-140
