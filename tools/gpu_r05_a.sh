#!/bin/bash
# round 5, call a: instruction-cache / two-stream micro-experiment, then the baseline stage table of the tree
OUT=gpurun_out/r05a; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 tools/micro/icache_persist > $OUT/icache_persist.txt 2>&1; echo "micro rc=$?"; cat $OUT/icache_persist.txt
timeout 200 python tools/bench_stages.py --rounds 7 base > $OUT/stages.txt 2>&1; tail -12 $OUT/stages.txt
timeout 60 python tools/gpu_state.py --compact > $OUT/gpu_state.txt 2>&1; tail -c 1500 $OUT/gpu_state.txt
