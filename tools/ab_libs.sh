#!/bin/bash
# A/B of library BUILDS on one box: tools/ab_libs.sh TAG "bench_stages args" libA.so libB.so ...
# (python -m matchering_amd.build --variant NAME -DFLAG makes tools/variants/libmgx_NAME.so).
# Alternates processes, three passes, so the box's clock state is shared by the variants.
OUT=gpurun_out/${1:-ab}; mkdir -p $OUT; ARGS=$2; shift 2
for pass in 1 2 3; do
  for lib in "$@"; do
    echo "== pass $pass $lib"
    MGX_LIB=$PWD/$lib timeout 200 python tools/bench_stages.py $ARGS base 2>&1 | tail -2
  done
done | tee $OUT/ab.txt
