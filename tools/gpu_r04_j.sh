#!/bin/bash
# round 4: k_analyze_thin (512 threads x 8 points, four radix-8 passes, next segment prefetched) -- parity tests, then
# the A/B against k_analyze<12> in one process, and the three-workgroup build (80 VGPRs, no scratch) against the
# four-workgroup one (64 VGPRs, 72 B of scratch)
OUT=gpurun_out/${1:-r04j}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hard_inputs.py -m gpu -q -x -p no:cacheprovider -k "analysis_stage or golden or near_tie or exact_tie or hard_material or degenerate" 2>&1 | tail -2
timeout 300 python tools/bench_stages.py --rounds 11 base nothin:MGX_ANALYZE_THIN=0 2>&1 | tail -4 | tee $OUT/thin_ab.txt
echo "== thin6"; MGX_LIB=$PWD/matchering_amd/libmgx_thin6.so timeout 300 python tools/bench_stages.py --rounds 11 base nothin:MGX_ANALYZE_THIN=0 2>&1 | tail -3 | tee -a $OUT/thin_ab.txt
