#!/bin/bash
# round 3: does warming again during the launch (every n-th workgroup of an XCD) keep the code in the L2?
OUT=gpurun_out/${1:-r03q}; mkdir -p $OUT; export TMPDIR=/tmp
python tools/gpu_state.py --compact > $OUT/gpu_state.json 2> $OUT/gpu_state.err; grep -A4 "112KiB" $OUT/gpu_state.json | tr -d '\n'; echo
for pass in 1 2; do
  for P in 0 16 64; do echo "== pass $pass MGX_CODE_WARM_PERIOD=$P"; MGX_CODE_WARM_PERIOD=$P timeout 200 python tools/bench_stages.py --rounds 7 base 2>&1 | tail -1; done
done | tee $OUT/ab_code_warm_period.txt
