#!/bin/bash
# round 4: the 8192-point transform on 512 threads in four passes (8 * 8 * 8 * 16, libmgx_fft13four.so) against
# 256 threads in three (16 * 16 * 32, the tree): parity of the variant, then the headline workload and fft_size 8192, A/B
OUT=gpurun_out/${1:-r04m}; mkdir -p $OUT; export TMPDIR=/tmp
MGX_LIB=$PWD/matchering_amd/libmgx_fft13four.so timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hard_inputs.py -m gpu -q -x -p no:cacheprovider > $OUT/pytest_variant.log 2>&1; echo "variant pytest rc=$?"; grep -E "passed|failed" $OUT/pytest_variant.log | tail -2
bash tools/ab_libs.sh ${1:-r04m}/headline "--rounds 7" matchering_amd/libmgx.so matchering_amd/libmgx_fft13four.so
for lib in libmgx.so libmgx_fft13four.so; do echo "== $lib fft_size 8192, 44.1 kHz, 4 min"; MGX_LIB=$PWD/matchering_amd/$lib timeout 200 python tools/bench_stages.py --rounds 5 --seconds 240 --fft-size 8192 base 2>&1 | tail -1; echo "== $lib 4min_x8"; done | tee $OUT/fft8192_ab.txt
for lib in libmgx.so libmgx_fft13four.so; do echo "== $lib bench"; MGX_LIB=$PWD/matchering_amd/$lib timeout 300 python bench.py 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['frac'], d['stage_ms'], d['parity'])"; done | tee $OUT/bench_ab.txt
