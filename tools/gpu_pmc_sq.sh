#!/bin/bash
# SQ counters (one pass, 8 slots) for the kernels matching KPAT.  Usage: bash tools/gpu_pmc_sq.sh TAG "c1 c2 ..." [workload]
TAG=$1; CNT=$2; WL=${3:-8min_full}
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d $OUT/pmc -o r -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-traffic --no-gpu-state --workload $WL > $OUT/pmc.log 2>&1
F=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
for C in $CNT; do python tools/pmc_summary.py $F $C | grep -i "${KPAT:-.}" | head -${KN:-5}; done
rm -rf $OUT/pmc
