#!/bin/bash
# round 4: the 16384-point transform on 1024 threads in four passes (16 * 8 * 8 * 16, the tree) against 512 threads in
# three (16 * 32 * 32, libmgx_fft14old.so): the whole GPU suite on the tree, then config #5, fft_size 8192 and 32768, A/B
OUT=gpurun_out/${1:-r04l}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -E "^FAILED" $OUT/pytest.log | head
for pass in 1 2 3; do for lib in libmgx_fft14old.so libmgx.so; do echo "== pass $pass $lib config 5"; MGX_LIB=$PWD/matchering_amd/$lib timeout 200 python tools/bench_stages.py --rounds 5 --seconds 240 --sample-rate 96000 --fft-size 16384 base 2>&1 | tail -1; done; done | tee $OUT/config5_ab.txt
for lib in libmgx_fft14old.so libmgx.so; do echo "== $lib fft_size 8192, 44.1 kHz, 4 min"; MGX_LIB=$PWD/matchering_amd/$lib timeout 200 python tools/bench_stages.py --rounds 5 --seconds 240 --fft-size 8192 base 2>&1 | tail -1; echo "== $lib fft_size 32768, 192 kHz, 2 min"; MGX_LIB=$PWD/matchering_amd/$lib timeout 200 python tools/bench_stages.py --rounds 5 --seconds 120 --sample-rate 192000 --fft-size 32768 base 2>&1 | tail -1; done | tee $OUT/other_sizes_ab.txt
