#!/bin/bash
# round 3: quiet-chunk path of the limiter -- parity first, then A/B; does a preheat change a box's class?
OUT=gpurun_out/${1:-r03f}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
python tools/gpu_state.py --compact > $OUT/gpu_state.json 2> $OUT/gpu_state.err; grep -E "shader_mhz|power_W|clocked_up|vbios" $OUT/gpu_state.json
timeout 900 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -6 $OUT/pytest.log | cut -c1-300
bash tools/ab_libs.sh ${1:-r03f} "--rounds 7" matchering_amd/libmgx.so matchering_amd/libmgx_noquiet.so matchering_amd/libmgx_ha6.so
mv $OUT/ab.txt $OUT/ab_limiter.txt
echo "== cold"; timeout 200 python tools/bench_stages.py --rounds 5 base 2>&1 | tail -2
echo "== preheated 2 s"; timeout 200 python tools/bench_stages.py --rounds 5 --preheat 2.0 base 2>&1 | tail -3
