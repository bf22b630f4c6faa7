#!/usr/bin/env python3
"""bench.py's file -> result legs alone (GPU box).  usage: [FTK_DEVICE_INFLATE=1] tools/e2e_legs.py [reps=5]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

torch.cuda.set_device(0)
res = bench.end_to_end(torch, reps=int(sys.argv[1]) if len(sys.argv) > 1 else 5)
res["FTK_DEVICE_INFLATE"] = os.environ.get("FTK_DEVICE_INFLATE", "0")
print(json.dumps(res))
