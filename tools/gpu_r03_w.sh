#!/bin/bash
# round 3, closing profiles: kernel table and SQ counters of the headline workload on the final code
OUT=gpurun_out/${1:-r03w}; mkdir -p $OUT; export TMPDIR=/tmp
python tools/gpu_state.py --compact > $OUT/gpu_state.json 2> $OUT/gpu_state.err; grep -A3 "112KiB" $OUT/gpu_state.json | tr -d '\n'; echo
WL=8min_full bash tools/gpu_variants.sh ${1:-r03w} "k_" base
KN=6 KPAT="k_" bash tools/gpu_pmc_sq.sh ${1:-r03w}/sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" 8min_full > $OUT/sq.txt 2>&1; head -50 $OUT/sq.txt | cut -c1-120
