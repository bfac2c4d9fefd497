#!/bin/bash
# round 3: parity of everything (with timeouts), state of the box, limiter / config #5 A/Bs
OUT=gpurun_out/${1:-r03d}; mkdir -p $OUT; export TMPDIR=/tmp
python tools/gpu_state.py --compact > $OUT/gpu_state.json 2> $OUT/gpu_state.err; cat $OUT/gpu_state.json
timeout 600 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log
bash tools/ab_libs.sh ${1:-r03d} "--rounds 7" matchering_amd/libmgx.so matchering_amd/libmgx_noskip.so matchering_amd/libmgx_ha6.so
mv $OUT/ab.txt $OUT/ab_limiter.txt
bash tools/ab_libs.sh ${1:-r03d} "--rounds 5 --seconds 240 --sample-rate 96000 --fft-size 16384" matchering_amd/libmgx.so matchering_amd/libmgx_conv13.so
mv $OUT/ab.txt $OUT/ab_config5.txt
