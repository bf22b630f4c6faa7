#!/bin/bash
# Kernel times inside the whole-genome text pass (tools/decode_pass.py run) under rocprofv3 --kernel-trace --stats.
# usage: tools/text_pass_profile.sh <out file under gpurun_out/>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
F=/tmp/ftk_sweep_genome.frag.gz
cd $GRAFT_REPO_ROOT
[ -f $F ] || python tools/decode_pass.py write $F 1 > /dev/null 2>&1
python tools/decode_pass.py run $F 5 2>/dev/null | tail -1 > $OUT.pass.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ftk_tp_prof -- python3 $GRAFT_REPO_ROOT/tools/decode_pass.py run $F 3 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py stats /tmp/ftk_tp_prof > $OUT
rm -rf /tmp/ftk_tp_prof
cat $OUT.pass.json; head -14 $OUT
