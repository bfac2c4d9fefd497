#!/bin/bash
# round 5, call d: the limiter instantiated for its window geometry, and the shadow launches, against round 4's library
OUT=gpurun_out/r05d; mkdir -p $OUT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "limiter or master_matches" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest.log
bash tools/ab_libs.sh r05d "--rounds 7" tools/variants/libmgx_r04.so matchering_amd/libmgx.so
timeout 400 python tools/bench_stages.py --rounds 7 base gen:MGX_LIMIT_GENERAL=1 sh1:MGX_SHADOW=1 sh4:MGX_SHADOW=4 sh2:MGX_SHADOW=2 sh3:MGX_SHADOW=3 sh6:MGX_SHADOW=6 > $OUT/stages.txt 2>&1; tail -12 $OUT/stages.txt
timeout 60 python -c "
import sys; sys.path.insert(0,'tools')
from gpu_state import compact_state; s=compact_state(); print(s['pci_bus'], s['memory_probe']['ns_per_instruction_112KiB_code'])" 2>&1 | tail -1
timeout 200 python -X faulthandler -m pytest tests/test_batch.py -m gpu -x -q -p no:cacheprovider --timeout 150 -k "album" > $OUT/pytest_album.log 2>&1; echo "album rc=$?"; tail -30 $OUT/pytest_album.log
