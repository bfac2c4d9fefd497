#!/bin/bash
# round 4: the delay-line convolution in its final form (pairs of bins, the older half of a window kept, the newer half asked
# for a block ahead): GPU parity of everything that touches it, phase times, kernel table and counter traffic of config #5
OUT=gpurun_out/${1:-r04w}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_hard_inputs.py tests/test_batch.py -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $OUT/pytest.log
MGX_LIB=$PWD/matchering_amd/libmgx_convphases.so timeout 300 python tools/conv_delay_phases.py 2>&1 | tee $OUT/phases.txt
WL=96k_16k_full bash tools/gpu_variants.sh ${1:-r04w} "k_" base | head -12; mv $OUT/kernel_stats_base.txt $OUT/kernel_stats_96k_16k_full.txt
bash tools/gpu_pmc.sh ${1:-r04w} 96k_16k_full > $OUT/pmc_config5.log 2>&1; grep "k_conv_delay" $OUT/pmc_FETCH_SIZE.txt $OUT/pmc_WRITE_SIZE.txt
for C in FETCH_SIZE WRITE_SIZE; do mv $OUT/pmc_$C.txt $OUT/pmc_${C}_96k_16k_full.txt; done
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; python -c "
import json; d=json.loads(open('$OUT/bench.json').readline()); print(d['ms_per_step'], d['roofline']['frac'], d['stage_ms'], {k: (v['ms_per_step'], v['frac_of_8TBs']) for k, v in d['other_workloads'].items()}, d['parity']['rms'])"
