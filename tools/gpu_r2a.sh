#!/bin/bash
# round-2 session a: GPU parity tests on the new limiter, limiter A/B, kernel stats
OUT=gpurun_out/r2a; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
python tools/bench_stages.py --rounds 9 v2:MGX_LIMITER=2 w4:MGX_LIMITER=3w4 w5:MGX_LIMITER=3w5 w6:MGX_LIMITER=3w6 w4p:MGX_LIMITER=3w4p w5p:MGX_LIMITER=3w5p w6p:MGX_LIMITER=3w6p > $OUT/ab_limiter.txt 2>&1; cat $OUT/ab_limiter.txt
