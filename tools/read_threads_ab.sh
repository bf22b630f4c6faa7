mkdir -p gpurun_out/r4h
for rt in 4 8; do
  FTK_READ_THREADS=$rt FTK_DECODE_TIMING=1 python tools/bam_big_run.py > gpurun_out/r4h/bam_big_rt$rt.json 2> gpurun_out/r4h/bam_big_rt$rt.err
  FTK_READ_THREADS=$rt FTK_E2E_REPS=4 python tools/e2e_genome_bench.py all 30 12 delfi > gpurun_out/r4h/genome_rt$rt.json 2>/dev/null
done
