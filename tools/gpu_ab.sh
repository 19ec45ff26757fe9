#!/bin/bash
# tools/gpu_ab.sh [bench args --] lib1.so lib2.so ...: force-only step time of several builds, alternating, 2 rounds (on the GPU box)
OUT=gpurun_out; mkdir -p $OUT
ARGS="--no-cpu-baseline --primary-only --steps 400 --warmup 40"
for rep in 1 2; do
  for v in "$@"; do
    NBNXM_HIP_LIB=$v timeout -k 10 200 python bench.py $ARGS > $OUT/ab.log 2>/dev/null || { echo "$v FAILED"; tail -3 $OUT/ab.log; exit 1; }
    echo "$(basename $v) $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab.log) $(grep -o '"k_calc_nb": [0-9.]*' $OUT/ab.log)"
  done
done
