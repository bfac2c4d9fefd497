#!/bin/bash
# round 4: packed f32 complex arithmetic (v_pk_add/mul/fma_f32, libmgx_packed.so = -DMGX_PACKED_MATH) against the tree:
# parity of the variant, then headline, config #5 and fft_size 8192, A/B
OUT=gpurun_out/${1:-r04s}; mkdir -p $OUT; export TMPDIR=/tmp
MGX_LIB=$PWD/matchering_amd/libmgx_packed.so timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hard_inputs.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_variant.log 2>&1; echo "variant pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_variant.log | tail -2; grep -E "^FAILED" $OUT/pytest_variant.log | head
bash tools/ab_libs.sh ${1:-r04s}/headline "--rounds 7" matchering_amd/libmgx.so matchering_amd/libmgx_packed.so
for pass in 1 2; do for lib in libmgx.so libmgx_packed.so; do echo "== pass $pass $lib config 5"; MGX_LIB=$PWD/matchering_amd/$lib timeout 200 python tools/bench_stages.py --rounds 5 --seconds 240 --sample-rate 96000 --fft-size 16384 base 2>&1 | tail -1; echo "== pass $pass $lib fft_size 8192, 44.1 kHz, 4 min"; MGX_LIB=$PWD/matchering_amd/$lib timeout 200 python tools/bench_stages.py --rounds 5 --seconds 240 --fft-size 8192 base 2>&1 | tail -1; done; done | tee $OUT/other_ab.txt
