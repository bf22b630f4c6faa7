#!/bin/bash
# What the GPU box offers a whole-genome BAM (BASELINE config 5): cores, memory, scratch space and its write / read rates.
# usage (from the repo root on the box): bash tools/box_probe.sh > gpurun_out/box_probe.txt
echo "cores: $(nproc)  (python: $(python3 -c 'import os; print(len(os.sched_getaffinity(0)))'))"
grep -m1 "model name" /proc/cpuinfo
free -g | head -2
df -h /tmp /dev/shm "$PWD" 2>/dev/null
T=${TMPDIR:-/tmp}/ftk_probe_$$
dd if=/dev/zero of=$T bs=4M count=1000 conv=fdatasync 2>&1 | tail -1
dd if=$T of=/dev/null bs=4M 2>&1 | tail -1
rm -f $T
