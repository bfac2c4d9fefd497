#!/bin/bash
# round 4: device handles (lanes) per GPU for the batch workload -- eight 4-minute pairs -- with the count forced (MGX_LANES)
OUT=gpurun_out/${1:-r04lanes}; mkdir -p $OUT; export TMPDIR=/tmp
for pass in 1 2; do for n in 2 3 4 5 6; do echo -n "pass $pass MGX_LANES=$n: "; MGX_LANES=$n timeout 200 python bench.py --workload 4min_x8_full --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-traffic --no-gpu-state 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['config'].get('lanes', d['config']))" ; done; done | tee $OUT/lanes.txt
