#!/bin/bash
# Force-only step time against the work shares of the age classes of a SIMD's waves (NBNXM_HIP_CLASS_SHARES5, oldest first)
OUT=gpurun_out; mkdir -p $OUT
for s in "$@"; do
  export NBNXM_HIP_CLASS_SHARES5=$s
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $OUT/cls.log 2>&1 || exit 1
  echo "$s $(grep -o '"ms_per_step": [0-9.]*' $OUT/cls.log) $(grep -o '"k_calc_nb": [0-9.]*' $OUT/cls.log) $(grep -o '"ms_per_gpu_resident_md_step": [0-9.]*' $OUT/cls.log)"
done
