#!/bin/bash
# round 3, closing: the whole GPU suite and the bench line with the driver's settings on the final code
OUT=gpurun_out/${1:-r03s}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log | cut -c1-300
timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json; tail -3 $OUT/bench.err
