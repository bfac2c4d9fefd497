#!/bin/bash
# round 4 profiles of the tree: rocprofv3 kernel tables (headline workload and config #5), counter traffic of the two
# streaming kernels, the bench line with the driver's settings, the self-launched two-rank line, and 4 / 8 self-launched
# ranks sharing the one GPU (the host side of the driver's 4- and 8-GPU runs; RCCL over loop-back sockets)
OUT=gpurun_out/${1:-r04p}; mkdir -p $OUT; export TMPDIR=/tmp
WL=8min_full bash tools/gpu_variants.sh ${1:-r04p} "k_" base; mv $OUT/kernel_stats_base.txt $OUT/kernel_stats_8min_full.txt
WL=96k_16k_full bash tools/gpu_variants.sh ${1:-r04p} "k_" base > /dev/null; mv $OUT/kernel_stats_base.txt $OUT/kernel_stats_96k_16k_full.txt
bash tools/gpu_pmc.sh ${1:-r04p} 8min_full > $OUT/pmc.log 2>&1; tail -12 $OUT/pmc.log | grep "k_limit\|k_conv<\|k_analyze"
for C in FETCH_SIZE WRITE_SIZE; do mv $OUT/pmc_$C.txt $OUT/pmc_${C}_8min_full.txt; done
bash tools/gpu_pmc.sh ${1:-r04p} 96k_16k_full > $OUT/pmc_config5.log 2>&1; tail -12 $OUT/pmc_config5.log | grep "k_conv"
for C in FETCH_SIZE WRITE_SIZE; do mv $OUT/pmc_$C.txt $OUT/pmc_${C}_96k_16k_full.txt; done
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-400 $OUT/bench.json
timeout 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-traffic --no-gpu-state > $OUT/bench_gpus2.json 2> $OUT/bench_gpus2.err; echo "bench --gpus 2 rc=$?"; cut -c1-300 $OUT/bench_gpus2.json; wc -l $OUT/bench_gpus2.json
for n in 4 8; do timeout 300 python bench.py --gpus $n --steps 5 --warmup 2 --workload 8min_fir_only --no-traffic --no-gpu-state --no-cpu-baseline > $OUT/bench_gpus$n.json 2> $OUT/bench_gpus$n.err; echo "bench --gpus $n rc=$?"; python -c "
import json,sys; d=json.loads(open('$OUT/bench_gpus$n.json').readline()); print(d['n_gpus'], d['launch'], d['value'], d['rccl_fir_allgather'], [round(x,4) for x in d['rank_seconds']])"; done
