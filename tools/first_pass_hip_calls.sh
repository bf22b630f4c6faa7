#!/bin/bash
# Which HIP calls a fresh process's FIRST text pass spends its time in (rocprofv3 --hip-runtime-trace --stats; no counters).
# usage: tools/first_pass_hip_calls.sh <out file under gpurun_out/>
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
F=/tmp/ftk_sweep_genome.frag.gz
cd $GRAFT_REPO_ROOT
[ -f $F ] || python tools/decode_pass.py write $F 1 > /dev/null 2>&1
FTK_DECODE_TIMING=1 python tools/decode_pass.py run $F 2 2> $OUT.timing.txt | tail -1 > $OUT.pass.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --hip-runtime-trace --stats --output-format csv -d /tmp/ftk_fp_prof -- python3 $GRAFT_REPO_ROOT/tools/decode_pass.py run $F 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/ftk_fp_prof -name "*hip_api_stats.csv" | head -1)
python - "$f" > $OUT <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("# rocprofv3 --hip-runtime-trace --stats: one fresh process, ONE whole-genome text pass")
for r in rows[:18]:
    print(f"{r['Name']:44s} calls {int(r['Calls']):6d}  total {float(r['TotalDurationNs'])/1e6:9.2f} ms  avg {float(r['AverageNs'])/1e3:9.1f} us  max {float(r['MaxNs'])/1e6:8.2f} ms")
PY
rm -rf /tmp/ftk_fp_prof
cat $OUT.pass.json; cat $OUT; grep -i "stream\|first\|ms" $OUT.timing.txt | head -30
