#!/bin/bash
OUT=gpurun_out/r05_s; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -X faulthandler -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider --timeout 300 -x -k "fft_size or unsupported or partitioned" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
echo "== 192 kHz, fft_size 65536, 120 s"
timeout 200 python tools/bench_stages.py --seconds 120 --sample-rate 192000 --fft-size 65536 base 2>&1 | tail -3 | tee $OUT/stages_65536.txt
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-traffic > $OUT/bench_headline.json 2> $OUT/bench.err; python - <<PY
import json
d = json.loads(open('$OUT/bench_headline.json').readline())
print(d['ms_per_step'], d['stage_ms'], d['gpu_state']['memory_probe']['ns_per_instruction_112KiB_code'])
PY
