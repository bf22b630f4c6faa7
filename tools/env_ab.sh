#!/bin/bash
# A/B of environment settings on one box and one file: alternating fresh processes of tools/first_pass_probe.py
# (N passes each; the first one of a process is its cold pass), every pass's total printed.
# usage: bash tools/env_ab.sh text|bam "VAR=a" "VAR=b" [rounds=3] [passes=6]     (a setting may hold several VAR=x words)
kind=${1:-text}; A=$2; B=$3; rounds=${4:-3}; passes=${5:-6}
export FTK_PROBE_DIR=${FTK_PROBE_DIR:-/tmp/ftk_env_ab}
for i in $(seq 1 $rounds); do
  for setting in "$A" "$B"; do
    echo -n "$setting: "
    env $setting python3 tools/first_pass_probe.py $kind $passes 2>&1 | grep "^pass" | sed -E 's/.*total ([0-9.]+) s.*/\1/' | tr '\n' ' '
    echo
  done
done
rm -rf "$FTK_PROBE_DIR"
