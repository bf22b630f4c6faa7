#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of libftk_hip.so's device code (hipcc -Rpass-analysis=
kernel-resource-usage on every .hip file), demangled.  usage: tools/resource_usage.py > profiles/rN_kernel_resource_usage.txt"""
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "finaletoolkit_amd", "csrc")
rows = []
for src in sorted(f for f in os.listdir(CSRC) if f.endswith(".hip")):
    r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{CSRC}",
                        "-Rpass-analysis=kernel-resource-usage", "--cuda-device-only", "-c", os.path.join(CSRC, src), "-o",
                        "/dev/null"], capture_output=True, text=True)
    cur = None
    for line in r.stderr.splitlines():
        m = re.search(r"remark: (.*) \[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = {"file": src, "name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.splitlines()
print("%-18s %-112s %5s %5s %4s %7s %7s" % ("file", "kernel", "VGPR", "SGPR", "occ", "LDS B", "scratch"))
for r, n in zip(rows, names):
    n = n.replace("(anonymous namespace)::", "")
    n = re.sub(r"\(.*", "", n).replace("ftk::", "")
    print("%-18s %-112s %5s %5s %4s %7s %7s" % (r["file"], n[:112], r.get("VGPRs", "?"), r.get("TotalSGPRs", "?"),
                                                 r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?"),
                                                 r.get("ScratchSize [bytes/lane]", "?")))
