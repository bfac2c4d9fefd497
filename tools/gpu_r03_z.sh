#!/bin/bash
# round 3: phase stamps of the level-correction kernels (trace variant of the library), then parity and stage times
OUT=gpurun_out/${1:-r03z}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
MGX_LIB=$PWD/matchering_amd/libmgx_tailtrace.so timeout 200 python tools/tail_trace.py > $OUT/tail_trace.txt 2>&1
tail -25 $OUT/tail_trace.txt | cut -c1-160
timeout 500 python -m pytest tests -m gpu -q -k "golden or scalars or correct or level or hard or rounds" -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -8
timeout 120 python tools/bench_stages.py --rounds 9 base > $OUT/stages.txt 2>&1; cat $OUT/stages.txt | cut -c1-150
