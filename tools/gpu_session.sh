#!/bin/bash
# One validation call on a GPU box: the whole GPU suite (with the slowest tests listed), the driver's bench line, the
# kernel table of the headline workload.   bash tools/gpu_session.sh [tag]
OUT=gpurun_out/${1:-session}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider --timeout 300 --durations=25 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -E "^FAILED|Timeout" $OUT/pytest.log | head; grep -E "s (call|setup)" $OUT/pytest.log | head -25
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -2 $OUT/bench.err; python - <<PY
import json
d = json.loads(open('$OUT/bench.json').readline())
print(d['ms_per_step'], d['value'], d['pipeline_hbm_model']['frac_of_8TBs'], d['pipeline_hbm_model'].get('measured_bytes'), d['pipeline_hbm_model'].get('measured_frac_of_8TBs'))
print(d['stage_ms'])
print('roofline', d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'], '| other', [(r['kernel'][:16], r['frac'], r['traffic']) for r in d['roofline_other']])
cb = d['cpu_baseline']; print('cpu', cb['kind'], cb['value'], cb['host'], 'port', cb.get('port', {}).get('value'), 'all cores', cb.get('all_cores'))
print({k: (v['ms_per_step'], v['frac_of_8TBs']) for k, v in d['other_workloads'].items()})
print(d['parity']['rms'], d['speedup_vs_cpu'], d['gpu_state']['memory_probe']['ns_per_instruction_112KiB_code'], d['rank_gpus'])
PY
