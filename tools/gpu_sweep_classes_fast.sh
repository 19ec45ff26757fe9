#!/bin/bash
# Force-only step time (primary figure only) against the work shares of the age classes (NBNXM_HIP_CLASS_SHARES5, oldest first)
OUT=gpurun_out; mkdir -p $OUT
for s in "$@"; do
  export NBNXM_HIP_CLASS_SHARES5=$s
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --primary-only --steps 300 --warmup 30 > $OUT/cls.log 2>/dev/null || exit 1
    echo "$s $(grep -o '"ms_per_step": [0-9.]*' $OUT/cls.log) $(grep -o '"k_calc_nb": [0-9.]*' $OUT/cls.log)"
  done
done
