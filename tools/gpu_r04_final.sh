#!/bin/bash
# round 4, last call: the bench line with the driver's settings, the headline workload's kernel table, then the whole GPU suite
OUT=gpurun_out/${1:-r04final}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$OUT/bench.json').readline()); print(d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['stage_ms'], {k: (v['ms_per_step'], v['frac_of_8TBs']) for k, v in d['other_workloads'].items()}, d['parity']['rms'], d['gpu_state']['memory_probe']['ns_per_instruction_112KiB_code'])"
WL=8min_full bash tools/gpu_variants.sh ${1:-r04final} "k_" base | head -8; mv $OUT/kernel_stats_base.txt $OUT/kernel_stats_8min_full.txt
timeout 330 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -E "^FAILED" $OUT/pytest.log | head
