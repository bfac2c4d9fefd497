#!/bin/bash
# round 4: the default filter (4096 taps) on N = 4F = 16384-point blocks (k_conv_wide, the tree) against N = 2F
# (k_conv<13,false>, MGX_NO_CONV_WIDE=1): parity of everything that convolves, then the headline workload A/B, config #5,
# and the phase times of both 16384-point kernels
OUT=gpurun_out/${1:-r04wide}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hard_inputs.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -1 $OUT/pytest.log
for pass in 1 2 3; do timeout 300 python tools/bench_stages.py --rounds 7 wide old:MGX_NO_CONV_WIDE=1 2>&1 | tail -2; done | tee $OUT/ab.txt
timeout 200 python tools/bench_stages.py --rounds 5 --seconds 240 --sample-rate 96000 --fft-size 16384 base 2>&1 | tail -1 | tee $OUT/config5.txt
MGX_LIB=$PWD/matchering_amd/libmgx_convphases.so timeout 200 python tools/conv_delay_phases.py --wide 2>&1 | tee $OUT/phases_wide.txt
MGX_LIB=$PWD/matchering_amd/libmgx_convphases.so timeout 200 python tools/conv_delay_phases.py 2>&1 | tee $OUT/phases_delay.txt
