#!/bin/bash
# A/B of bench.py step shapes on ONE box (boxes differ by +-10 %): alternates the variants given as KEY=VALUE env
# settings, e.g.  tools/bench_ab.sh FTK_BENCH_MERGED=0 FTK_BENCH_MERGED=1
mkdir -p gpurun_out
for rep in 1 2; do
  for v in "$@"; do
    env "$v" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end > gpurun_out/ab.json 2> gpurun_out/ab.err
    python - "$v" <<'PY'
import json, sys
j = json.loads(open("gpurun_out/ab.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], "ms/step", j["ms_per_step"], "dominant", r["kernel"], r["frac"], r["avg_launch_ms"], "second",
      (r.get("second_kernel") or {}).get("avg_launch_ms"), "whole", r["whole_step"]["frac"], j["checks"])
PY
  done
done
