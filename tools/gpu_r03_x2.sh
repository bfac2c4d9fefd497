#!/bin/bash
# round 3: where does k_fir_taps_fft spend its time -- rocprof kernel durations, unrolled against rolled loops
OUT=gpurun_out/${1:-r03x2}; mkdir -p $OUT; export TMPDIR=/tmp
for V in base tapsu1; do
  if [ "$V" = base ]; then LIB=$PWD/matchering_amd/libmgx.so; else LIB=$PWD/matchering_amd/libmgx_$V.so; fi
  for CFG in "--seconds 240 --sample-rate 96000 --fft-size 16384" "--seconds 480"; do
    MGX_LIB=$LIB timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python tools/bench_stages.py --rounds 6 $CFG base > $OUT/prof.log 2>&1
    DB=$(find $OUT/prof -name "*.db" | head -1)
    python tools/rocprof_stats.py $DB > $OUT/ks.txt 2>&1
    echo "== $V $CFG"; grep -E "k_fir|k_match_curve" $OUT/ks.txt | cut -c1-40,75-140
    rm -rf $OUT/prof
  done
done
