#!/bin/bash
# A/B a library environment switch on one box: bash tools/ab_env.sh VAR VALUE   (alternates unset / set)
export TMPDIR=/tmp
VAR=$1; VAL=$2
for W in 0 1 0 1; do
  if [ $W = 1 ]; then export $VAR=$VAL; else unset $VAR; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$VAR=' + ('$VAL' if $W else 'unset'), 'ms/step', d['ms_per_step'], 'conv ms', d['roofline']['kernel_ms'])"
done
