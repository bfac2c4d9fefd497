#!/bin/bash
OUT=gpurun_out/r05_q; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -m gpu -q -p no:cacheprovider --timeout 300 -x \
  -k "fft_size or factored or long_fir or golden or 96k or analysis_stage or hard" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
echo "== config 5: factored operator / curve tiles against round 4's"
timeout 200 python tools/bench_stages.py --seconds 240 --sample-rate 96000 --fft-size 16384 base round4:MGX_FIR_ROUND4=1 tile32:MGX_CURVE_TILE32=1 both:MGX_FIR_ROUND4=1,MGX_CURVE_TILE32=1 2>&1 | tail -12 | tee $OUT/ab_fir_chain_config5.txt
echo "== 192 kHz, fft_size 65536 and 32768, 120 s"
timeout 200 python tools/bench_stages.py --seconds 120 --sample-rate 192000 --fft-size 65536 base 2>&1 | tail -3 | tee $OUT/stages_65536.txt
timeout 200 python tools/bench_stages.py --seconds 120 --sample-rate 192000 --fft-size 32768 base round4:MGX_FIR_ROUND4=1 2>&1 | tail -4 | tee $OUT/stages_32768.txt
timeout 100 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-traffic --no-gpu-state --workload 96k_16k_full > $OUT/prof.log 2>&1
python tools/rocprof_stats.py $(find $OUT/prof -name "*.db" | head -1) > $OUT/kernel_stats_96k_16k_full.txt 2>&1; head -16 $OUT/kernel_stats_96k_16k_full.txt; tail -1 $OUT/prof.log | cut -c1-200; rm -rf $OUT/prof
