#!/bin/bash
# quick correctness probe of the lane-parallel inflate loop under a timeout (a wave that waits for the other forever must not
# take the box down): one small text image, then the suite.  usage: bash tools/lanes_quick.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export FTK_INFLATE_LANES=1
timeout 120 python3 $R/tools/inflate_bench.py 21 2>&1 | tail -4 || echo "TIMEOUT / FAIL in inflate_bench"
