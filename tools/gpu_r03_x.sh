#!/bin/bash
# round 3: tap synthesis by a split transform (k_fir_taps_sub + k_fir_taps_combine) against the cosine sum
OUT=gpurun_out/${1:-r03x}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 400 python -m pytest tests -m gpu -q -k "golden or scalars or 96k or fft or hard or fir" -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -5
for CFG in "--seconds 240 --sample-rate 96000 --fft-size 16384" "--seconds 240 --sample-rate 96000 --fft-size 8192" "--seconds 120 --sample-rate 96000 --fft-size 32768"; do
  echo "== $CFG" >> $OUT/tap_synthesis.txt
  timeout 120 python tools/bench_stages.py --rounds 7 $CFG base cos:MGX_TAPS_BY_COSINE_SUM=1 >> $OUT/tap_synthesis.txt 2>&1
done
cut -c1-150 $OUT/tap_synthesis.txt
