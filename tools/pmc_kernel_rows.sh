#!/bin/bash
# SQ counters of the kernels of tools/kernel_rows.py (next-row kernels or the BAM-mode kernels): where their cycles go.
# usage (repo root on the GPU box): bash tools/pmc_kernel_rows.sh <out_dir under gpurun_out/> [next|bam]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
W=${2:-next}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAIT_INST_LDS"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set --output-format csv -d $OUT/sq_$tag -- python3 $GRAFT_REPO_ROOT/tools/kernel_rows.py $W 2 > /dev/null 2> $OUT/sq_$tag.err
done
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py pmc $OUT/sq_* > $OUT/pmc_sq_${W}.txt
rm -rf $OUT/sq_SQ_*
grep -v "^#" $OUT/pmc_sq_${W}.txt | grep -i "cleav\|gc_count\|adjust\|feat\|wps" | head -80
