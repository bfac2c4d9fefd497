#!/bin/bash
# round 4: cache policy of the convolution's stores (non-temporal since round 1) -- do the mid plane (85 MB, read back by
# round 0 of the level correction) and the frames (169 MB, read back by the limiter) come from the Infinity Cache when
# they are stored with the default policy?
OUT=gpurun_out/${1:-r04i}; mkdir -p $OUT; export TMPDIR=/tmp
bash tools/ab_libs.sh ${1:-r04i} "--rounds 9" matchering_amd/libmgx.so matchering_amd/libmgx_mid0.so matchering_amd/libmgx_mid1.so matchering_amd/libmgx_y0.so matchering_amd/libmgx_both0.so
