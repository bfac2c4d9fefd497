#!/bin/bash
OUT=gpurun_out/${1:-r03l}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 200 python tools/box_class.py $OUT > $OUT/box.log 2>&1; echo "rc=$?"; tail -5 $OUT/box.log
