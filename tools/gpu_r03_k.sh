#!/bin/bash
# round 3: where do kernel arguments live?  HIP_FORCE_DEV_KERNARG unset / 0 / 1 on the same box
OUT=gpurun_out/${1:-r03k}; mkdir -p $OUT; export TMPDIR=/tmp
python tools/gpu_state.py --compact > $OUT/gpu_state.json 2> $OUT/gpu_state.err; grep -E "shader_mhz_all|clocked_up|vbios" $OUT/gpu_state.json
for pass in 1 2; do
for K in unset 0 1; do
  echo "== pass $pass HIP_FORCE_DEV_KERNARG=$K"
  if [ "$K" = unset ]; then env -u HIP_FORCE_DEV_KERNARG timeout 200 python tools/bench_stages.py --rounds 7 base 2>&1 | tail -1
  else HIP_FORCE_DEV_KERNARG=$K timeout 200 python tools/bench_stages.py --rounds 7 base 2>&1 | tail -1; fi
done; done | tee $OUT/kernarg.txt
