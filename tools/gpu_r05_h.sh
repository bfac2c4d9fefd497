#!/bin/bash
# round 5, call h: the whole GPU suite on the tree with chunk = workgroup number; persistent static grid A/B; a bench line
OUT=gpurun_out/r05h; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests -m gpu -q -p no:cacheprovider --timeout 200 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -E "^FAILED|Timeout" $OUT/pytest.log | head
timeout 300 python tools/bench_stages.py --rounds 7 base tick:MGX_LIMIT_TICKETS=1 gen:MGX_LIMIT_GENERAL=1 > $OUT/stages.txt 2>&1; tail -4 $OUT/stages.txt
bash tools/ab_libs.sh r05h "--rounds 7" matchering_amd/libmgx.so tools/variants/libmgx_persist.so 2>&1 | grep -E "^==|^base"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --no-traffic > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$OUT/bench.json').readline()); print(d['ms_per_step'], d['value'], d['pipeline_hbm_model']['frac_of_8TBs'], d['stage_ms'], d['roofline']['kernel'], d['roofline']['frac'], d['gpu_state']['memory_probe']['ns_per_instruction_112KiB_code'])"
