#!/bin/bash
# round 4: config #5's convolution as a frequency-domain delay line (k_conv_delay) against the partitioned kernel
# (MGX_NO_CONV_DELAY=1): parity first, then the A/B on the 96 kHz / 16384-tap workload, then chunk variants
OUT=gpurun_out/${1:-r04n}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "delay_line or partitioned or long_fir or 96k or convolution" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
for pass in 1 2; do echo "== pass $pass"; timeout 300 python tools/bench_stages.py --rounds 5 --seconds 240 --sample-rate 96000 --fft-size 16384 base old:MGX_NO_CONV_DELAY=1 2>&1 | tail -3; done | tee $OUT/config5_ab.txt
for lib in libmgx.so libmgx_cdch2.so libmgx_cdnopre.so; do echo "== $lib"; MGX_LIB=$PWD/matchering_amd/$lib timeout 200 python tools/bench_stages.py --rounds 5 --seconds 240 --sample-rate 96000 --fft-size 16384 base 2>&1 | tail -1; done | tee $OUT/chunk_variants.txt
