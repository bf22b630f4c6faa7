cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for v in new; do
  if [ $v = old ]; then export FTK_LIB=$R/finaletoolkit_amd/libftk_old.so; fi
  for t in inflate_bench bam_inflate_probe; do
    rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/ab_${v}_$t -- python3 $R/tools/$t.py > $R/gpurun_out/ab_${v}_$t.log 2>&1
    echo "== $v $t"; python3 $R/tools/prof_summary.py $R/gpurun_out/ab_${v}_$t 2>/dev/null | grep -i "bgzf" | head -4
  done
done
