#!/bin/bash
OUT=gpurun_out/${1:-session}; mkdir -p $OUT; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $OUT/pytest.log
python tools/bench_stages.py --rounds 9 base > $OUT/stages_full.txt 2>&1; cat $OUT/stages_full.txt
python tools/bench_stages.py --rounds 9 --fir-only base > $OUT/stages_fir.txt 2>&1; tail -3 $OUT/stages_fir.txt
rocprofv3 --kernel-trace --stats -d $OUT/prof -o r -- python tools/bench_stages.py --rounds 5 base > $OUT/prof.log 2>&1
DB=$(find $OUT/prof -name "*.db" | head -1)
python tools/rocprof_stats.py $DB > $OUT/kernel_stats.txt 2>&1; head -30 $OUT/kernel_stats.txt
python tools/rocprof_timeline.py $DB > $OUT/timeline.txt 2>&1; tail -20 $OUT/timeline.txt
rm -rf $OUT/prof
python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json; tail -3 $OUT/bench.err
