#!/bin/bash
# round 5, call e: does the instruction-fetch micro-experiment CAUSE the slow state of a box?  probe, micro, probe, stages, probe;
# then the limiter's phase profile on the geometry-specialised kernel; then tests/test_batch.py with a stack dump if it hangs
OUT=gpurun_out/r05e; mkdir -p $OUT; export TMPDIR=/tmp
probe() { timeout 60 python -c "
import sys; sys.path.insert(0,'tools')
import mgx_probe
m = mgx_probe.memory(0); print('$1', 'cold', round(m[5], 2), 'again', round(m[6], 2))" 2>&1 | tail -1; }
probe first
timeout 120 tools/micro/icache_persist > $OUT/icache_persist.txt 2>&1; head -3 $OUT/icache_persist.txt | tail -2
probe after_micro
timeout 200 python tools/bench_stages.py --rounds 5 base > $OUT/stages.txt 2>&1; tail -2 $OUT/stages.txt
probe after_stages
MGX_LIB=$PWD/tools/variants/libmgx_phases.so timeout 200 python tools/limiter_phases.py > $OUT/limiter_phases.txt 2>&1; head -32 $OUT/limiter_phases.txt
timeout 400 python -X faulthandler -m pytest tests/test_batch.py -m gpu -x -q -p no:cacheprovider --timeout 120 > $OUT/pytest_batch.log 2>&1; echo "batch rc=$?"; tail -40 $OUT/pytest_batch.log | cut -c1-300
