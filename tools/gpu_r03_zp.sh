#!/bin/bash
# round 3, closing profiles of the final tree: kernel tables of the headline workload and of config #5
OUT=gpurun_out/${1:-r03zp}; mkdir -p $OUT; export TMPDIR=/tmp
python tools/gpu_state.py --compact > $OUT/gpu_state.json 2> $OUT/gpu_state.err; grep -A3 "112KiB" $OUT/gpu_state.json | tr -d '\n'; echo
WL=8min_full bash tools/gpu_variants.sh ${1:-r03zp} "k_" base; mv $OUT/kernel_stats_base.txt $OUT/kernel_stats_8min_full.txt
WL=96k_16k_full bash tools/gpu_variants.sh ${1:-r03zp} "k_" base; mv $OUT/kernel_stats_base.txt $OUT/kernel_stats_96k_16k_full.txt
