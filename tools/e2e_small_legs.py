#!/usr/bin/env python3
"""bench.py's end-to-end legs without the whole-genome one (seconds instead of a minute): for stream timelines,
`bash tools/trace_run.sh small tools/e2e_small_legs.py` then `python tools/trace_view.py gpurun_out/trace_small`."""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["FTK_BENCH_GENOME_E2E"] = "0"
import torch  # noqa: E402
import bench  # noqa: E402

r = bench.end_to_end(torch, reps=3)
print(json.dumps({k: v["total_s"] for k, v in r.items() if isinstance(v, dict)}))
