#!/bin/bash
# adjust_median row of tools/kernel_rows.py for experiment builds of the library (finaletoolkit_amd/libftk_cv_<name>.so)
for v in "$@"; do
  echo $v
  FTK_LIB=$PWD/finaletoolkit_amd/libftk_cv_$v.so python tools/kernel_rows.py next 5 2>/dev/null | python -c "
import json,sys
d=json.load(sys.stdin); d=d.get('next_rows', d); v=d['adjust_median_kernel']; print(v['frac'], v['avg_launch_ms'], v['best_launch_ms'])"
done
