#!/bin/bash
# The round's record call: GPU suite, the driver's bench line, kernel tables (rocprofv3 --kernel-trace --stats) of the
# headline and config #5 workloads, FETCH_SIZE / WRITE_SIZE tables.   bash tools/gpu_final.sh TAG
TAG=${1:-final}; OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/gpu_session.sh $TAG
for WL in 8min_full 96k_16k_full; do
  timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/prof_$WL -o r -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-traffic --no-gpu-state --workload $WL > $OUT/prof_$WL.log 2>&1
  DB=$(find $OUT/prof_$WL -name "*.db" | head -1)
  python tools/rocprof_stats.py $DB > $OUT/kernel_stats_$WL.txt 2>&1
  echo "== $WL"; head -12 $OUT/kernel_stats_$WL.txt; tail -1 $OUT/prof_$WL.log | cut -c1-300
  rm -rf $OUT/prof_$WL
done
bash tools/gpu_pmc.sh $TAG 8min_full > $OUT/pmc.log 2>&1; grep -E "k_limit|k_conv_wide<|k_analyze|k_correction_round" $OUT/pmc_*.txt | head -12
