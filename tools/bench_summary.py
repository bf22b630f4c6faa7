#!/usr/bin/env python3
"""A bench line (bench.py's JSON) as a short table: headline, roofline, kernel rows, every file -> result leg with its floor.
usage: tools/bench_summary.py gpurun_out/bench.json"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"value {d['value']} {d['unit']}  ms_per_step {d['ms_per_step']}  n_gpus {d['n_gpus']}  checks {all(d['checks'].values())}")
print(f"roofline {r['kernel']}: {r['avg_launch_ms'] * 1e3:.1f} us per launch, frac {r['frac']}, traffic x{(r['traffic'] or 0) / r['algorithmic_bytes_per_launch']:.3f}, whole step frac {r['whole_step']['frac']}")
for k, v in (r.get("bam_kernels") or {}).items():
    if isinstance(v, dict):
        print(f"  bam {k:32s} frac {v['frac']}  at_survey_bytes {v.get('frac_at_survey_bytes')}  {v['avg_launch_ms'] * 1e3:.0f} us")
for k, v in (d.get("next_rows") or {}).items():
    if isinstance(v, dict):
        print(f"  next {k:31s} frac {v.get('frac')}  {v.get('avg_launch_ms', 0) * 1e3:.0f} us")
c = d.get("cpu_baseline") or {}
print("cpu_baseline", c.get("value"), c.get("unit"), "cores", c.get("cores"), "| all cores", (c.get("all_cores") or {}).get("value"),
      "| reference-shaped python", (c.get("reference_shaped_python") or {}).get("value"))


def leg(name, v, pad=""):
    f = v.get("floor") or {}
    print(f"{pad}{name:32s} first {v.get('first_s')}  median {v.get('median_s')}  best {v.get('best_s')}  floor {f.get('floor_s')}  "
          f"of floor {f.get('frac_of_floor_best')} / {f.get('frac_of_floor_median')}  ok {v.get('results_ok')}")


e = d.get("end_to_end") or {}
for k, v in e.items():
    if isinstance(v, dict) and "best_s" in v:
        leg(k, v)
    elif k == "commands" and isinstance(v, dict):
        for kk, vv in v.items():
            if isinstance(vv, dict):
                leg(kk, vv, "  cmd ")
            else:
                print("  cmd", kk, vv)
    elif isinstance(v, dict):
        print(k, {x: v[x] for x in list(v)[:6]})
if e.get("error"):
    print("ERROR", e["error"])
