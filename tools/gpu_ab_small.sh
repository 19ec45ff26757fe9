#!/bin/bash
# tools/gpu_ab_small.sh lib1.so lib2.so ...: force-only step of the 3k-atom and the 96k-atom box for several builds, alternating
OUT=gpurun_out; mkdir -p $OUT
for rep in 1 2; do
  for a in 3k 96k; do
    for v in "$@"; do
      NBNXM_HIP_LIB=$v timeout -k 10 200 python bench.py --atoms $a --no-cpu-baseline --primary-only --steps 600 --warmup 60 > $OUT/ab.log 2>/dev/null || { echo "$v FAILED"; exit 1; }
      echo "$a $(basename $v) $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab.log)"
    done
  done
done
