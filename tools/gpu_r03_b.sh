#!/bin/bash
# round 3, second session: parity of the changed kernels, limiter A/B (quiet skip, attack forget), config #5 A/B + profiles
OUT=gpurun_out/${1:-r03b}; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
python tools/gpu_state.py > $OUT/gpu_state.json 2> $OUT/gpu_state.err; grep -A2 '"loaded"\|idle_one' $OUT/gpu_state.json | grep mhz
bash tools/ab_libs.sh ${1:-r03b} "--rounds 7" matchering_amd/libmgx.so matchering_amd/libmgx_noskip.so matchering_amd/libmgx_ha6.so
mv $OUT/ab.txt $OUT/ab_limiter.txt
bash tools/ab_libs.sh ${1:-r03b} "--rounds 5 --seconds 240 --sample-rate 96000 --fft-size 16384" matchering_amd/libmgx.so matchering_amd/libmgx_conv13.so
mv $OUT/ab.txt $OUT/ab_config5.txt
WL=96k_16k_full bash tools/gpu_variants.sh ${1:-r03b} "k_conv\|k_fir_taps\|k_analyze\|k_limit" base conv13
bash tools/gpu_pmc.sh ${1:-r03b} 96k_16k_full > $OUT/pmc_config5.log 2>&1; grep -A6 "^kernel" $OUT/pmc_FETCH_SIZE.txt | head -8; grep -A6 "^kernel" $OUT/pmc_WRITE_SIZE.txt | head -8
KPAT="k_conv" bash tools/gpu_pmc_sq.sh ${1:-r03b}/sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS" 96k_16k_full > $OUT/sq_config5.txt 2>&1; cat $OUT/sq_config5.txt
