#!/bin/bash
# round 3: limiter at 5 / 6 workgroups per CU (sh parked in the plane, reload at the store), kernel table of the headline workload
OUT=gpurun_out/${1:-r03g}; mkdir -p $OUT; export TMPDIR=/tmp
python tools/gpu_state.py --compact > $OUT/gpu_state.json 2> $OUT/gpu_state.err; grep -E "shader_mhz|power_W|clocked_up" $OUT/gpu_state.json
bash tools/ab_libs.sh ${1:-r03g} "--rounds 7" matchering_amd/libmgx.so matchering_amd/libmgx_late.so matchering_amd/libmgx_w5.so matchering_amd/libmgx_w5e.so matchering_amd/libmgx_w6.so
mv $OUT/ab.txt $OUT/ab_limiter_wgs.txt
WL=8min_full bash tools/gpu_variants.sh ${1:-r03g} "k_" base
