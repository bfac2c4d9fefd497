#!/bin/bash
# Time kernels of experimental library variants (libmgx_<name>.so built with
#   python -m matchering_amd.build --variant <name> <flags>) under rocprofv3.
# Usage: bash tools/gpu_variants.sh TAG KERNEL_SUBSTRING name1 name2 ...   ("base" = the product library)
TAG=$1; KPAT=$2; shift 2
OUT=gpurun_out/$TAG; mkdir -p $OUT; export TMPDIR=/tmp
for V in "$@"; do
  if [ "$V" = base ]; then LIB=$PWD/matchering_amd/libmgx.so; else LIB=$PWD/tools/variants/libmgx_$V.so; fi
  MGX_LIB=$LIB timeout 240 rocprofv3 --kernel-trace --stats -d $OUT/prof_$V -o r -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-secondary --no-traffic --no-gpu-state --workload ${WL:-8min_full} > $OUT/prof_$V.log 2>&1
  DB=$(find $OUT/prof_$V -name "*.db" | head -1)
  python tools/rocprof_stats.py $DB > $OUT/kernel_stats_$V.txt 2>&1
  echo "== $V"; grep -i "$KPAT" $OUT/kernel_stats_$V.txt
  rm -rf $OUT/prof_$V
done
