#!/bin/bash
# One gpurun call: GPU parity tests, bench, A/B of a library variant, rocprofv3 kernel trace.
# Usage (from the repo root on the GPU box):  bash tools/gpu_session.sh TAG [variant]
TAG=${1:-x}
VARIANT=$2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log; tail -5 $OUT/pytest.log
python bench.py --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json
if [ -n "$VARIANT" ]; then
  MGX_LIB=$PWD/matchering_amd/libmgx_$VARIANT.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_$VARIANT.json 2> $OUT/bench_$VARIANT.err
  echo "variant $VARIANT:"; cat $OUT/bench_$VARIANT.json
fi
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --workload 8min_full > $OUT/prof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
python tools/rocprof_stats.py $DB > $OUT/kernel_stats.txt 2>&1; cat $OUT/kernel_stats.txt
if [ -n "$VARIANT" ]; then
  MGX_LIB=$PWD/matchering_amd/libmgx_$VARIANT.so rocprofv3 --kernel-trace --stats -d $OUT/prof_$VARIANT -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --workload 8min_full > $OUT/prof_$VARIANT.log 2>&1
  DB=$(find $OUT/prof_$VARIANT -name "*.db" | head -1)
  python tools/rocprof_stats.py $DB > $OUT/kernel_stats_$VARIANT.txt 2>&1; echo "variant $VARIANT"; head -12 $OUT/kernel_stats_$VARIANT.txt
fi
