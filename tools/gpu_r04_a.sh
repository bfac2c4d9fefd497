#!/bin/bash
# round 4, first call: the GPU suite on the new tree (probes outside libmgx, mgx_comm_count), the self-launched
# two-rank bench on this one-GPU box (RCCL over loop-back, ranks_seen), and the default bench line
OUT=gpurun_out/${1:-r04a}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log | cut -c1-300
timeout 400 python bench.py --gpus 2 --steps 10 --warmup 3 --no-traffic --no-gpu-state > $OUT/bench_gpus2.json 2> $OUT/bench_gpus2.err; echo "bench --gpus 2 rc=$?"; cut -c1-1500 $OUT/bench_gpus2.json; tail -5 $OUT/bench_gpus2.err
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cut -c1-600 $OUT/bench.json; tail -3 $OUT/bench.err
