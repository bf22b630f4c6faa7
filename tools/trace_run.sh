#!/bin/bash
# Kernel + memory-copy + HIP-API trace of one Python tool, for tools/trace_view.py.
#   gpurun -- 'bash tools/trace_run.sh bam tools/bam_e2e_bench.py'   ->  gpurun_out/trace_bam/
#   python tools/trace_view.py gpurun_out/trace_bam 12
# (no --pmc here: counters go in their own runs, tools/profile_round.sh)
name=$1; shift
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
script=$(realpath "$1"); shift
cd /tmp && export TMPDIR=/tmp
rm -rf "$R/gpurun_out/trace_$name"
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d "$R/gpurun_out/trace_$name" -- python3 "$script" "$@" > "$R/gpurun_out/trace_$name.log" 2>&1
tail -2 "$R/gpurun_out/trace_$name.log" | cut -c1-400
