#!/bin/bash
# round 4: analysis with half the row traffic through LDS (A/B against the previous build), the N = 4F convolution
# plan (MGX_CONV_WIDE=1) against N = 2F, then the analysis and small-fft tests on the new build
OUT=gpurun_out/${1:-r04g}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "analysis_stage or golden or other_fft or 32768" 2>&1 | tail -2
bash tools/ab_libs.sh ${1:-r04g} "--rounds 9" matchering_amd/libmgx_base.so matchering_amd/libmgx.so
timeout 300 python tools/bench_stages.py --rounds 9 base wide:MGX_CONV_WIDE=1 2>&1 | tail -4 | tee $OUT/conv_wide.txt
