#!/bin/bash
# round 3: the whole GPU suite, the bench line end to end, config #5 profiles (kernel table, FETCH / WRITE, SQ)
OUT=gpurun_out/${1:-r03e}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -8 $OUT/pytest.log | cut -c1-300
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-1500 $OUT/bench.json; tail -3 $OUT/bench.err
WL=96k_16k_full bash tools/gpu_variants.sh ${1:-r03e} "k_conv\|k_fir\|k_analyze\|k_limit\|k_match\|k_corr" base
bash tools/gpu_pmc.sh ${1:-r03e} 96k_16k_full > $OUT/pmc_config5.log 2>&1; grep -A7 "^kernel" $OUT/pmc_FETCH_SIZE.txt | head -9; grep -A7 "^kernel" $OUT/pmc_WRITE_SIZE.txt | head -9
KPAT="k_conv" bash tools/gpu_pmc_sq.sh ${1:-r03e}/sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" 96k_16k_full > $OUT/sq_config5.txt 2>&1; cat $OUT/sq_config5.txt
