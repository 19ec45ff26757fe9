#!/bin/bash
OUT=gpurun_out; mkdir -p $OUT
for rep in 1 2; do for mode in tree head forced; do
  unset NBNXM_HIP_LIB NBNXM_HIP_DIAGNOSTICS NBNXM_HIP_CLASS_SHARES5
  [ $mode = head ] && export NBNXM_HIP_LIB=variants/head.so
  [ $mode = forced ] && export NBNXM_HIP_LIB=variants/head.so NBNXM_HIP_DIAGNOSTICS=1 NBNXM_HIP_CLASS_SHARES5=1350,1185,1024,865,696
  timeout -k 10 200 python bench.py --atoms 1m --no-cpu-baseline --primary-only --steps 300 --warmup 30 --condition-steps 500 > $OUT/cls.log 2>/dev/null || exit 1
  echo "1m $mode $(grep -o '"ms_per_step": [0-9.]*' $OUT/cls.log)"
done; done
