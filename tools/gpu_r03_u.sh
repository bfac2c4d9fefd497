#!/bin/bash
# round 3: fft_size 16384 analysed as two 8192-point transforms per segment (two workgroups per CU) vs one 16384-point transform
OUT=gpurun_out/${1:-r03u}; mkdir -p $OUT; export TMPDIR=/tmp
for pass in 1 2; do
  for S in 0 1; do echo "== pass $pass MGX_ANALYZE_SPLIT_14=$S"; MGX_ANALYZE_SPLIT_14=$S timeout 200 python tools/bench_stages.py --rounds 5 --seconds 240 --sample-rate 96000 --fft-size 16384 base 2>&1 | tail -1; done
done | tee $OUT/ab_split14.txt
MGX_ANALYZE_SPLIT_14=1 timeout 400 python -m pytest tests -m gpu -q -k "96k or analysis_stage or fft_32768" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log | cut -c1-200
