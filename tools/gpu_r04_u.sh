#!/bin/bash
# round 4: k_conv_delay's multiply phase on PAIRS of bins (one filter fetch for a bin and its mirror: the tree, chunks of
# 4 pairs; libmgx_cdchp2.so: chunks of 2 pairs) against the previous form (libmgx_cdprev.so: every thread its own row's
# 16 bins, chunks of 2 bins); parity first, phase times last
OUT=gpurun_out/${1:-r04u}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "delay_line or long_fir or 96k or partitioned" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
MGX_LIB=$PWD/matchering_amd/libmgx_cdchp2.so timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "delay_line or long_fir" > $OUT/pytest_chp2.log 2>&1; echo "chp2 pytest rc=$?"; tail -1 $OUT/pytest_chp2.log
for pass in 1 2 3; do for lib in libmgx_cdprev.so libmgx.so libmgx_cdchp2.so; do echo "== pass $pass $lib"; MGX_LIB=$PWD/matchering_amd/$lib timeout 200 python tools/bench_stages.py --rounds 5 --seconds 240 --sample-rate 96000 --fft-size 16384 base 2>&1 | tail -1; done; done | tee $OUT/variants.txt
for lib in libmgx_convphases.so libmgx_convphases2.so; do echo "== $lib"; MGX_LIB=$PWD/matchering_amd/$lib timeout 300 python tools/conv_delay_phases.py 2>&1; done | tee $OUT/phases.txt
