#!/bin/bash
# round 4, second call: (1) the new error-path tests; (2) variant builds checked against the oracle where they
# differ; (3) alternating A/Bs on this one box: DPP scans in the limiter, the analysis prefetch at 3 and 4
# workgroups per CU, round 0 of the level correction on 1024 / 1536 / 2048 workgroups.
OUT=gpurun_out/${1:-r04b}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_device_errors.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "correction_step_counts" 2>&1 | tail -2
echo "== dpp: limiter tests"; MGX_LIB=$PWD/matchering_amd/libmgx_dpp.so timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_limiter_kat.py -m gpu -q -p no:cacheprovider -k "limiter or golden" 2>&1 | tail -2
for v in anl3p anl4p; do echo "== $v: analysis tests"; MGX_LIB=$PWD/matchering_amd/libmgx_$v.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "analysis_stage or golden or other_fft" 2>&1 | tail -2; done
bash tools/ab_libs.sh ${1:-r04b} "--rounds 7" matchering_amd/libmgx.so matchering_amd/libmgx_dpp.so matchering_amd/libmgx_anl3p.so matchering_amd/libmgx_anl3.so matchering_amd/libmgx_anl4p.so
echo "== round-0 workgroups"; timeout 300 python tools/bench_stages.py --rounds 9 base w1536:MGX_ROUND_WGS=1536 w2048:MGX_ROUND_WGS=2048 w3072:MGX_ROUND_WGS=3072 notail:MGX_NO_TAIL=1 2>&1 | tail -8 | tee $OUT/round_wgs.txt
