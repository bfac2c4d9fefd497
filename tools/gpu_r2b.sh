#!/bin/bash
# round-2 session b: limiter A/B with the FULL fast path + ablations, SQ counters of old vs new
OUT=gpurun_out/r2b; mkdir -p $OUT; export TMPDIR=/tmp
python tools/bench_stages.py --rounds 9 v2:MGX_LIMITER=2 w4:MGX_LIMITER=3w4 w5:MGX_LIMITER=3w5 w6:MGX_LIMITER=3w6 w4p:MGX_LIMITER=3w4p nopoll:MGX_LIMITER=3w4a1 copy8:MGX_LIMITER=3w4a2 copy4:MGX_LIMITER=3w4a2,MGX_LIM_LDS_PAD=20000 copy6:MGX_LIMITER=3w4a2,MGX_LIM_LDS_PAD=6000 > $OUT/ab_limiter.txt 2>&1; cat $OUT/ab_limiter.txt
for PASS in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM"; do
  TAG=$(echo $PASS | cut -d' ' -f1)
  rocprofv3 --pmc $PASS --kernel-trace --output-format csv -d $OUT/pmc_$TAG -o r -- python tools/bench_stages.py --rounds 2 v2:MGX_LIMITER=2 w4:MGX_LIMITER=3w4 > $OUT/pmc_$TAG.log 2>&1
  F=$(find $OUT/pmc_$TAG -name "*counter_collection.csv" | head -1)
  python tools/pmc_table.py $F k_limit > $OUT/pmc_$TAG.txt; cat $OUT/pmc_$TAG.txt
  rm -rf $OUT/pmc_$TAG
done
