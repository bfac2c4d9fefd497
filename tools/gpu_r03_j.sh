#!/bin/bash
# round 3: limiter with the release word published before the backward attack pass (limit_chunk_full) vs the old order
OUT=gpurun_out/${1:-r03j}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $OUT/smoke.log
timeout 600 python -m pytest tests -m gpu -q -k "limiter or golden or full_size or hard or device_error" > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log | cut -c1-300
bash tools/ab_libs.sh ${1:-r03j} "--rounds 7" matchering_amd/libmgx.so matchering_amd/libmgx_oldorder.so
mv $OUT/ab.txt $OUT/ab_limiter_order.txt
python tools/gpu_state.py --compact > $OUT/gpu_state.json 2> $OUT/gpu_state.err; grep -E "shader_mhz|vbios" $OUT/gpu_state.json
