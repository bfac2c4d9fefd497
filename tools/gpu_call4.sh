#!/bin/bash
OUT=gpurun_out/r05_r; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout 300 -x -k "factored or long_fir or golden" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
echo "== headline: curve tiles of 16 bins"
timeout 200 python tools/bench_stages.py --rounds 9 base tile16:MGX_CURVE_TILE16=1 base2 tile16b:MGX_CURVE_TILE16=1 2>&1 | tail -8 | tee $OUT/ab_curve_tile16_headline.txt
echo "== config 5"
timeout 200 python tools/bench_stages.py --seconds 240 --sample-rate 96000 --fft-size 16384 base tile16:MGX_CURVE_TILE16=1 round4:MGX_FIR_ROUND4=1,MGX_CURVE_TILE32=1 2>&1 | tail -6 | tee $OUT/ab_config5.txt
timeout 100 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-traffic --no-gpu-state --workload 96k_16k_full > $OUT/prof.log 2>&1
python tools/rocprof_stats.py $(find $OUT/prof -name "*.db" | head -1) > $OUT/kernel_stats_96k_16k_full.txt 2>&1; grep -E "fir_apply|match_curve|taps" $OUT/kernel_stats_96k_16k_full.txt; grep -o '"ms_per_step": [0-9.]*' $OUT/prof.log | head -2; rm -rf $OUT/prof
