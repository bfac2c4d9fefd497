#!/bin/bash
# round 4, third call: the persistent limiter (next chunk fetched under the current one) -- the whole GPU suite on it,
# then an alternating A/B against the one-shot kernel of round 3 in one process
OUT=gpurun_out/${1:-r04c}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 700 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log | cut -c1-300
timeout 300 python tools/bench_stages.py --rounds 11 base oneshot:MGX_LIMIT_ONESHOT=1 2>&1 | tail -5 | tee $OUT/ab_persistent.txt
timeout 300 python tools/bench_stages.py --rounds 7 --seconds 240 base oneshot:MGX_LIMIT_ONESHOT=1 2>&1 | tail -3 | tee -a $OUT/ab_persistent.txt
