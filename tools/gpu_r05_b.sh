#!/bin/bash
# round 5, call b: what the first pass over a kernel's code costs -- a stage's launches twice in a row
OUT=gpurun_out/r05b; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python tools/bench_stages.py --rounds 7 base conv2:MGX_DEV_REPEAT_CONV=2 lim2:MGX_DEV_REPEAT_LIMIT=2 lim3:MGX_DEV_REPEAT_LIMIT=3 > $OUT/stages.txt 2>&1; tail -12 $OUT/stages.txt
timeout 60 python -c "
import sys; sys.path.insert(0,'tools')
from gpu_state import compact_state; s=compact_state(); print(s['pci_bus'], s['memory_probe']['ns_per_instruction_112KiB_code'])" 2>&1 | tail -1
