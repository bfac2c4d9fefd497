#!/bin/bash
OUT=gpurun_out/r2c; mkdir -p $OUT; export TMPDIR=/tmp
python tools/bench_stages.py --rounds 9 v2:MGX_LIMITER=2 w4:MGX_LIMITER=3w4 p4:MGX_LIMITER=3p4 p3:MGX_LIMITER=3p3 p5:MGX_LIMITER=3p5 p4p:MGX_LIMITER=3p4p p4nopoll:MGX_LIMITER=3p4a1 p4copy8:MGX_LIMITER=3p4a2 p4copy4:MGX_LIMITER=3p4a2,MGX_LIM_LDS_PAD=20000 > $OUT/ab_limiter.txt 2>&1; cat $OUT/ab_limiter.txt
