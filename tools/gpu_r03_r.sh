#!/bin/bash
# round 3: instruction cache counters of the big kernels
OUT=gpurun_out/${1:-r03r}; mkdir -p $OUT; export TMPDIR=/tmp
python tools/gpu_state.py --compact > $OUT/gpu_state.json 2> $OUT/gpu_state.err; grep -A4 "112KiB" $OUT/gpu_state.json | tr -d '\n'; echo
rocprofv3 -L 2>/dev/null | grep -i -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u | tr '\n' ' '; echo
KN=12 KPAT="k_" bash tools/gpu_pmc_sq.sh ${1:-r03r}/sq "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAVES" 8min_full > $OUT/icache.txt 2>&1; cat $OUT/icache.txt | cut -c1-140
