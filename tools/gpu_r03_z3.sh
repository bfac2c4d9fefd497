#!/bin/bash
# round 3: round 0's decision moved into k_correction_tail's deciding workgroup -- parity and stage times
OUT=gpurun_out/${1:-r03z3}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 500 python -m pytest tests -m gpu -q -k "golden or scalars or correct or level or hard or rounds" -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -8
timeout 120 python tools/bench_stages.py --rounds 11 base > $OUT/stages.txt 2>&1; cat $OUT/stages.txt | cut -c1-150
