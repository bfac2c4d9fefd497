#!/bin/bash
# HBM traffic per kernel launch from the L2's memory-side counters, one rocprofv3 pass per counter
# (MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE do not fit one pass; counters get their own run).
# Usage: bash tools/gpu_pmc.sh TAG [workload]
TAG=$1; WL=${2:-8min_full}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 240 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $OUT/pmc_$C -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-traffic --no-gpu-state --workload $WL > $OUT/pmc_$C.log 2>&1
  F=$(find $OUT/pmc_$C -name "*counter_collection.csv" | head -1)
  echo "== $C ($F)"; head -2 $F
  python tools/pmc_summary.py $F $C > $OUT/pmc_$C.txt; cat $OUT/pmc_$C.txt
  rm -rf $OUT/pmc_$C
done
