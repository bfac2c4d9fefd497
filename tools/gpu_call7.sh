#!/bin/bash
OUT=gpurun_out/r05_v; mkdir -p $OUT; export TMPDIR=/tmp
timeout 600 python -X faulthandler -m pytest tests/test_batch.py tests/test_device_errors.py -m gpu -q -p no:cacheprovider --timeout 300 -x > $OUT/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $OUT/pytest.log
(cd /tmp && timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 /root/repo/bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > /root/repo/$OUT/bench_torchrun_gpus2.json 2> /root/repo/$OUT/bench_torchrun_gpus2.err; echo "torchrun rc=$?")
tail -2 $OUT/bench_torchrun_gpus2.err | cut -c1-300; cut -c1-900 $OUT/bench_torchrun_gpus2.json
timeout 300 python bench.py --steps 10 --warmup 3 --workload 4min_x8_full --no-cpu-baseline --no-secondary --no-traffic --no-gpu-state > $OUT/bench_4min_x8.json 2> $OUT/bench_4min_x8.err; echo "lanes rc=$?"; grep -o '"ms_per_step": [0-9.]*' $OUT/bench_4min_x8.json | head -1; grep -o '"workload": "[^"]*"' $OUT/bench_4min_x8.json | head -1
