#!/bin/bash
# round 4, closing: smoke, the whole GPU suite (with durations), a soak of mixed shapes / configurations / lanes
OUT=gpurun_out/${1:-r04z}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $OUT/pytest.log | tail -2; grep -A9 "slowest" $OUT/pytest.log | tail -9
timeout 200 python tools/soak.py 60 2>&1 | tail -4 | tee $OUT/soak.txt
