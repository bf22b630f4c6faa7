#!/bin/bash
# motif pass of tools/kernel_rows.py against the chunk walker's blocks per CU (FTK_FEAT_BPC)
for b in ${BPCS:-32 16 8 4 2}; do
  echo "BPC $b"
  FTK_FEAT_BPC=$b python tools/kernel_rows.py next 5 2>/tmp/kr.err | python -c "
import json,sys
d=json.load(sys.stdin)
d=d.get('next_rows', d)
v=d['motif_pass']; print(v.get('frac'), v.get('avg_launch_ms'), v.get('best_launch_ms'))
" || tail -5 /tmp/kr.err
done
