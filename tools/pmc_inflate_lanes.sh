#!/bin/bash
# SQ counters of both symbol loops of the device inflate on one chip-filling text image (tools/inflate_bench.py <contig>),
# one --pmc pass per counter.  usage: tools/pmc_inflate_lanes.sh <out_dir under gpurun_out/> [contig=1]
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
C=${2:-1}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export FTK_INFLATE_VECTOR_MATCHES=0
for lanes in ${LANES_LIST:-0 1}; do
  export FTK_INFLATE_LANES=$lanes
  for c in SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_LDS_BANK_CONFLICT SQ_INSTS_BRANCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/sq_$c -- python3 $GRAFT_REPO_ROOT/tools/inflate_bench.py $C > /dev/null 2> $OUT/sq_$c.err
  done
  ( cd $GRAFT_REPO_ROOT && python tools/prof_summary.py pmc $OUT/sq_* > $OUT/pmc_sq_inflate_lanes$lanes.txt )
  rm -rf $OUT/sq_SQ_*
  echo "== lanes=$lanes"; grep -i "inflate" $OUT/pmc_sq_inflate_lanes$lanes.txt
done
