#!/bin/bash
# round 5, call c: the error-word / requeue rework and the new bench line on the GPU
OUT=gpurun_out/r05c; mkdir -p $OUT; export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_device_errors.py tests/test_abi.py tests/test_batch.py -m gpu -x -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
timeout 500 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<'PY'
import json
d = json.loads(open('gpurun_out/r05c/bench.json').readline())
print(d['ms_per_step'], d['value'], d['stage_ms'])
print('roofline', d['roofline']['kernel'], d['roofline']['frac'], d['roofline']['traffic'])
print('other', [(r['kernel'], r['frac'], r['traffic']) for r in d['roofline_other']])
print('pipeline', d['pipeline_hbm_model'])
cb = d['cpu_baseline']; print('cpu', cb['kind'], cb['value'], cb.get('host'), cb.get('port', {}).get('value'), cb.get('all_cores'))
print('rank_gpus', d['rank_gpus'], d['gpu_state']['memory_probe']['ns_per_instruction_112KiB_code'], d['parity']['rms'])
PY
timeout 300 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/bench2.json 2> $OUT/bench2.err; echo "bench2 rc=$?"; python -c "
import json; d=json.loads(open('gpurun_out/r05c/bench2.json').readline()); print({k: d.get(k) for k in ('n_gpus','rank_gpus','physical_gpus','shared_gpu','n1_same_workload','value','rccl_fir_allgather')}, d['config'].get('lane_choice'), d['pipeline_hbm_model']['peak_GBs'])"
