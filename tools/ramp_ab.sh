#!/bin/bash
# A/B of the streams' short first reads (FTK_STREAM_RAMP) on one box and one file: alternating runs of
# tools/first_pass_probe.py, each a fresh process of N passes; prints every pass's total.
# usage: bash tools/ramp_ab.sh text|bam [rounds=3] [passes=6]
kind=${1:-text}; rounds=${2:-3}; passes=${3:-6}
export FTK_PROBE_DIR=${FTK_PROBE_DIR:-/tmp/ftk_ramp_ab}
for i in $(seq 1 $rounds); do
  for r in 0 4194304; do
    echo -n "ramp $r: "
    FTK_STREAM_RAMP=$r python3 tools/first_pass_probe.py $kind $passes 2>&1 | grep "^pass" | sed -E 's/.*total ([0-9.]+) s.*/\1/' | tr '\n' ' '
    echo
  done
done
rm -rf "$FTK_PROBE_DIR"
