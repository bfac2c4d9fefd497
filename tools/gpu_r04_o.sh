#!/bin/bash
# round 4: wave-local passes (no workgroup barrier between the passes after pass 0: the tree) against a barrier per pass
# (libmgx_passbar.so): the whole GPU suite on the tree first, then headline, config #5 and fft_size 8192, A/B
OUT=gpurun_out/${1:-r04o}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -E "^FAILED" $OUT/pytest.log | head
bash tools/ab_libs.sh ${1:-r04o}/headline "--rounds 7" matchering_amd/libmgx_passbar.so matchering_amd/libmgx.so
for pass in 1 2; do for lib in libmgx_passbar.so libmgx.so; do echo "== pass $pass $lib config 5"; MGX_LIB=$PWD/matchering_amd/$lib timeout 200 python tools/bench_stages.py --rounds 5 --seconds 240 --sample-rate 96000 --fft-size 16384 base 2>&1 | tail -1; echo "== pass $pass $lib fft_size 8192, 44.1 kHz, 4 min"; MGX_LIB=$PWD/matchering_amd/$lib timeout 200 python tools/bench_stages.py --rounds 5 --seconds 240 --fft-size 8192 base 2>&1 | tail -1; done; done | tee $OUT/other_ab.txt
