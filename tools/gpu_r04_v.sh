#!/bin/bash
# round 4: k_conv_delay on pairs (the tree: chunks of 2 pairs) against chunks of 1 pair (libmgx_cdchp1.so), the older half of
# a window kept in registers (libmgx_cdhold1.so) and that plus the newer half asked for a block early (libmgx_cdhold2.so)
OUT=gpurun_out/${1:-r04v}; mkdir -p $OUT; export TMPDIR=/tmp
for lib in libmgx_cdchp1.so libmgx_cdhold1.so libmgx_cdhold2.so; do MGX_LIB=$PWD/matchering_amd/$lib timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "delay_line or long_fir" > $OUT/pytest_$lib.log 2>&1; echo "$lib pytest rc=$?"; tail -1 $OUT/pytest_$lib.log; done
for pass in 1 2 3; do for lib in libmgx.so libmgx_cdchp1.so libmgx_cdhold1.so libmgx_cdhold2.so; do echo "== pass $pass $lib"; MGX_LIB=$PWD/matchering_amd/$lib timeout 200 python tools/bench_stages.py --rounds 5 --seconds 240 --sample-rate 96000 --fft-size 16384 base 2>&1 | tail -1; done; done | tee $OUT/variants.txt
