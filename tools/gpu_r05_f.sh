#!/bin/bash
# round 5, call f: every limiter chunk's life (entry, loaded, end, kind, compute unit) -- what bounds the kernel now?
OUT=gpurun_out/r05f; mkdir -p $OUT; export TMPDIR=/tmp
MGX_LIB=$PWD/tools/variants/libmgx_phases.so timeout 200 python tools/limiter_life.py > $OUT/limiter_life.txt 2>&1; cat $OUT/limiter_life.txt
timeout 200 python tools/bench_stages.py --rounds 7 base gen:MGX_LIMIT_GENERAL=1 > $OUT/stages.txt 2>&1; tail -3 $OUT/stages.txt
