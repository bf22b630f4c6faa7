cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $R/gpurun_out/prof_bam2 -- python3 $R/tools/bam_e2e_bench.py > $R/gpurun_out/prof_bam2.log 2>&1
ls -la $R/gpurun_out/prof_bam2/*/
