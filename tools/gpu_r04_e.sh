#!/bin/bash
# round 4: the whole GPU suite on the tree with the small-fft kernels, the non-finite input check, warm_code from the
# dispatch packet; then stage times (is anything slower than in r04a?)
OUT=gpurun_out/${1:-r04e}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|error" $OUT/pytest.log | tail -3; grep -E "^FAILED|^ERROR" $OUT/pytest.log | head -20
timeout 300 python tools/bench_stages.py --rounds 9 base nowarm:MGX_NO_CODE_WARM=1 2>&1 | tail -3 | tee $OUT/stages.txt
