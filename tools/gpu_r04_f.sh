#!/bin/bash
# which of the new GPU tests are slow?
OUT=gpurun_out/${1:-r04f}; mkdir -p $OUT; export TMPDIR=/tmp
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "other_fft_sizes or correction_step" --durations=12 --timeout 120 2>&1 | tail -22 | tee $OUT/durations_a.txt
timeout 500 python -m pytest tests/test_gpu_hard_inputs.py -m gpu -q -p no:cacheprovider -k "non_finite" --durations=6 --timeout 120 2>&1 | tail -14 | tee $OUT/durations_b.txt
