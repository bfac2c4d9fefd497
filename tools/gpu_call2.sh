#!/bin/bash
# validation of the factored FIR operator and fft_size 65536 + A/B of the analysis touch variants
OUT=gpurun_out/r05_p; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -p no:cacheprovider --timeout 300 -x \
  -k "fft_size or factored or partitioned or long_fir or golden or 96k or unsupported" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $OUT/pytest.log
echo "== config 5: factored operator against round 4's dense one"
timeout 200 python tools/bench_stages.py --seconds 240 --sample-rate 96000 --fft-size 16384 base round4:MGX_FIR_ROUND4=1 2>&1 | tail -12 | tee $OUT/ab_fir_factored.txt
echo "== config 5: analysis touch"
bash tools/ab_libs.sh r05_p_touch14 "--seconds 240 --sample-rate 96000 --fft-size 16384" matchering_amd/libmgx.so tools/variants/libmgx_touch14.so 2>&1 | grep -E "==|analyze|total|wall" | head -40
echo "== headline: analysis touch"
bash tools/ab_libs.sh r05_p_touch12 "" matchering_amd/libmgx.so tools/variants/libmgx_touch12.so 2>&1 | grep -E "==|analyze|total|wall" | head -40
