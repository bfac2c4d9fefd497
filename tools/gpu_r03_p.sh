#!/bin/bash
# round 3: code warming (the kernels' first workgroups read their own code as data) on / off, box class first
OUT=gpurun_out/${1:-r03p}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
python tools/gpu_state.py --compact > $OUT/gpu_state.json 2> $OUT/gpu_state.err; grep -A4 "112KiB" $OUT/gpu_state.json | tr -d '\n'; echo
for pass in 1 2 3; do
  echo "== pass $pass warm"; timeout 200 python tools/bench_stages.py --rounds 7 base 2>&1 | tail -1
  echo "== pass $pass MGX_NO_CODE_WARM=1"; MGX_NO_CODE_WARM=1 timeout 200 python tools/bench_stages.py --rounds 7 base 2>&1 | tail -1
done | tee $OUT/ab_code_warm.txt
