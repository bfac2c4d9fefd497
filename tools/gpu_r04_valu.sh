#!/bin/bash
# round 4, the last GPU minute: VALU wave-instructions per launch (SQ_INSTS_VALU) of the headline workload's kernels on the final tree
OUT=gpurun_out/${1:-r04valu}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 90 rocprofv3 --pmc SQ_INSTS_VALU --kernel-trace --output-format csv -d $OUT/pmc -o r -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-traffic --no-gpu-state --workload 8min_full > $OUT/pmc.log 2>&1
F=$(find $OUT/pmc -name "*counter_collection.csv" | head -1)
python tools/pmc_summary.py $F SQ_INSTS_VALU | head -8 | tee $OUT/sq_insts_valu.txt
rm -rf $OUT/pmc
