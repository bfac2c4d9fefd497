#!/bin/bash
# round 4: k_conv_delay, the newer half of the next window asked for behind the multiply (the tree, 48 B of scratch) against
# behind the inverse middle passes (libmgx_cdlate.so, 24 B)
OUT=gpurun_out/${1:-r04x}; mkdir -p $OUT; export TMPDIR=/tmp
MGX_LIB=$PWD/matchering_amd/libmgx_cdlate.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "delay_line or long_fir" > $OUT/pytest_late.log 2>&1; echo "late pytest rc=$?"; tail -1 $OUT/pytest_late.log
for pass in 1 2 3; do for lib in libmgx.so libmgx_cdlate.so; do echo "== pass $pass $lib"; MGX_LIB=$PWD/matchering_amd/$lib timeout 200 python tools/bench_stages.py --rounds 5 --seconds 240 --sample-rate 96000 --fft-size 16384 base 2>&1 | tail -1; done; done | tee $OUT/variants.txt
