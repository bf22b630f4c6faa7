#!/bin/bash
# Multi-rank logic of bench.py on a 1-GPU box: N ranks share GPU 0 and exchange through gloo (host copies).
# usage: tools/bench_2rank_1gpu.sh [N=2] [extra bench.py args]
N=${1:-2}; shift
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 WORLD_SIZE=$N LOCAL_RANK=0 FTK_BENCH_DIST_BACKEND=gloo
pids=()
for r in $(seq 1 $((N-1))); do RANK=$r python bench.py --gpus $N --no-cpu-baseline "$@" > /dev/null 2> gpurun_out/rank$r.err & pids+=($!); done
RANK=0 timeout 600 python bench.py --gpus $N --no-cpu-baseline "$@"
rc=$?
for p in "${pids[@]}"; do wait $p || rc=$?; done
exit $rc
