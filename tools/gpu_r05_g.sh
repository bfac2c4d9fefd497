#!/bin/bash
# round 5, call g: chunk = workgroup number (no ticket) and a persistent grid, against the committed limiter
OUT=gpurun_out/r05g; mkdir -p $OUT; export TMPDIR=/tmp
for v in blockorder persist; do
  MGX_LIB=$PWD/tools/variants/libmgx_$v.so timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "limiter or master_matches" 2>&1 | tail -1
done
bash tools/ab_libs.sh r05g "--rounds 7" matchering_amd/libmgx.so tools/variants/libmgx_blockorder.so tools/variants/libmgx_persist.so 2>&1 | grep -E "^==|^base"
MGX_LIB=$PWD/tools/variants/libmgx_bophases.so timeout 200 python tools/limiter_life.py > $OUT/limiter_life_blockorder.txt 2>&1; cat $OUT/limiter_life_blockorder.txt
