#!/bin/bash
# round 3, first session: what state is this box in, base stage times, -ffp-contract A/B, lanes sweep
OUT=gpurun_out/${1:-r03a}; mkdir -p $OUT; export TMPDIR=/tmp
python tools/gpu_state.py > $OUT/gpu_state.json 2> $OUT/gpu_state.err; cat $OUT/gpu_state.json | head -120
(rocm-smi -a > $OUT/rocm_smi_a.txt 2>&1; amd-smi static > $OUT/amd_smi_static.txt 2>&1; amd-smi metric > $OUT/amd_smi_metric.txt 2>&1) &
python tools/bench_stages.py --rounds 9 base > $OUT/stages_full.txt 2>&1; cat $OUT/stages_full.txt
wait
bash tools/ab_libs.sh ${1:-r03a} "--rounds 7" matchering_amd/libmgx.so matchering_amd/libmgx_fc.so
python tools/lanes_sweep.py > $OUT/lanes.txt 2>&1; cat $OUT/lanes.txt
python tools/gpu_state.py > $OUT/gpu_state_after.json 2>> $OUT/gpu_state.err; grep -A3 '"loaded"\|idle_one' $OUT/gpu_state_after.json
