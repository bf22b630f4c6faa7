#!/usr/bin/env python3
"""Print the per-repetition totals of a tools/e2e_genome_bench.py log or a tools/bam_e2e_bench.py JSON."""
import json
import sys

for line in open(sys.argv[1]):
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    if "reps" in d:
        print([r.get("total_s") for r in d["reps"]])
    elif "rep0" in d:
        print({k: (v["end_to_end_s"], v["waiting_for_contigs_s"]) for k, v in d.items() if k.startswith("rep")})
