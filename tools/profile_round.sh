#!/bin/bash
# Round profile collection on the GPU box (run from the repo root): kernel stats of bench.py, the two HBM PMC passes,
# SQ counters of the window-feature kernels (standalone, chained launches), kernel stats of the file -> result legs.
# usage: tools/profile_round.sh <out_dir under gpurun_out/>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-kernel-rows"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $B --steps 3 --warmup 1 --no-cpu-baseline --no-end-to-end > $OUT/bench_under_profiler.json 2> $OUT/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $B --steps 1 --warmup 0 --no-cpu-baseline --no-end-to-end > /dev/null 2> $OUT/pmc_write.err
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD; do
  KBENCH=feat rocprofv3 --pmc $c --output-format csv -d $OUT/sq_$c -- python3 $GRAFT_REPO_ROOT/tools/kbench.py 243199373 3 > /dev/null 2> $OUT/sq_$c.err
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/e2e_stats -- python3 $GRAFT_REPO_ROOT/tools/e2e_legs.py 3 > $OUT/e2e_under_profiler.json 2> $OUT/e2e_stats.err
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/genome_stats -- python3 $GRAFT_REPO_ROOT/tools/e2e_genome_bench.py all 30 12 delfi > $OUT/genome_leg_under_profiler.json 2> /dev/null
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py stats $OUT/genome_stats > $OUT/genome_leg_kernel_stats.txt
python tools/prof_summary.py stats $OUT/stats > $OUT/kernel_stats.txt
python tools/prof_summary.py pmc $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_hbm.txt
python tools/prof_summary.py pmc $OUT/sq_* > $OUT/pmc_sq_feature_kernels.txt
python tools/prof_summary.py stats $OUT/e2e_stats > $OUT/e2e_kernel_stats.txt
FTK_BENCH_FORCE_DIST=1 $B --no-cpu-baseline --no-end-to-end > $OUT/bench_force_dist.json 2> $OUT/bench_force_dist.err
KBENCH=feat python tools/kbench.py > $OUT/kbench_feat_fast.txt 2>&1
FTK_BENCH_MERGED=0 $B --no-cpu-baseline --no-end-to-end > $OUT/bench_two_launches.json 2> /dev/null
FTK_FEAT_FAST=0 KBENCH=feat python tools/kbench.py > $OUT/kbench_feat_general.txt 2>&1
FTK_BENCH_DETAIL=1 $B --no-cpu-baseline --no-end-to-end > $OUT/bench_detail.json 2> $OUT/bench_detail.err
rm -rf $OUT/stats $OUT/pmc_fetch $OUT/pmc_write $OUT/sq_SQ_* $OUT/e2e_stats $OUT/genome_stats
# ---- round 5: the kernels behind the headline launch (tools/kernel_rows.py) under the profiler ------------------
# BAM-mode kernels of config 5 (chr1-sized 60x contig with read1 columns): kernel stats + the two HBM counter passes
cd /tmp
K="python3 $GRAFT_REPO_ROOT/tools/kernel_rows.py"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bam_stats -- $K bam 5 > $OUT/bam_kernels_under_profiler.json 2> $OUT/bam_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/bam_fetch -- $K bam 2 > /dev/null 2> $OUT/bam_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/bam_write -- $K bam 2 > /dev/null 2> $OUT/bam_write.err
# next-row kernels (cleavage, motif pass, adjust median, G + C count)
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/next_stats -- $K next 5 > $OUT/next_rows_under_profiler.json 2> $OUT/next_stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/next_fetch -- $K next 2 > /dev/null 2> $OUT/next_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/next_write -- $K next 2 > /dev/null 2> $OUT/next_write.err
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py stats $OUT/bam_stats > $OUT/bam_kernel_stats.txt
python tools/prof_summary.py pmc $OUT/bam_fetch $OUT/bam_write > $OUT/bam_pmc_hbm.txt
python tools/prof_summary.py stats $OUT/next_stats > $OUT/nextrow_kernel_stats.txt
python tools/prof_summary.py pmc $OUT/next_fetch $OUT/next_write > $OUT/nextrow_pmc_hbm.txt
$K all 5 > $OUT/kernel_rows.json 2> $OUT/kernel_rows.err
rm -rf $OUT/bam_stats $OUT/bam_fetch $OUT/bam_write $OUT/next_stats $OUT/next_fetch $OUT/next_write
ls $OUT
