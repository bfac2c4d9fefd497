#!/bin/bash
# round 4: the next kernel's code read into the L2s from the tail of the current one (warm_next) against a build without
# it, three alternating passes; limiter and analysis tests first (quiet-chunk threshold, warm_next reads code memory)
OUT=gpurun_out/${1:-r04h}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_limiter_kat.py -m gpu -q -x -p no:cacheprovider -k "limiter or golden or other_fft or correction" 2>&1 | tail -2
bash tools/ab_libs.sh ${1:-r04h} "--rounds 9" matchering_amd/libmgx_nonext.so matchering_amd/libmgx.so
echo "== config 5"; for lib in libmgx_nonext.so libmgx.so; do MGX_LIB=$PWD/matchering_amd/$lib timeout 200 python tools/bench_stages.py --rounds 5 --seconds 240 --sample-rate 96000 --fft-size 16384 base 2>&1 | tail -1; done | tee $OUT/config5.txt
