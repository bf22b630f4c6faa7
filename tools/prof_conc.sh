cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/inflate_concurrency_probe.py 2>&1 | tail -1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/prof_conc -- python3 $R/tools/inflate_concurrency_probe.py > $R/gpurun_out/prof_conc.log 2>&1
tail -1 $R/gpurun_out/prof_conc.log
