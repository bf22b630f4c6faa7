#!/bin/bash
# SQ counters of the device inflate kernels on tools/inflate_bench.py (one --pmc pass per counter).  usage: tools/pmc_inflate.sh <out_dir under gpurun_out/>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH; do
  rocprofv3 --pmc $c --output-format csv -d $OUT/sq_$c -- python3 $GRAFT_REPO_ROOT/tools/inflate_bench.py 21 > /dev/null 2> $OUT/sq_$c.err
done
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py pmc $OUT/sq_* > $OUT/pmc_sq_inflate.txt
rm -rf $OUT/sq_SQ_*
grep -i "inflate\|crc" $OUT/pmc_sq_inflate.txt
