#!/bin/bash
# Force step time against the work shares of the five age classes (NBNXM_HIP_CLASS_SHARES5, oldest first; diagnostics)
OUT=gpurun_out; mkdir -p $OUT
ATOMS=${ATOMS:-96k}
export NBNXM_HIP_DIAGNOSTICS=1
for rep in 1 2; do
for s in "$@"; do
  export NBNXM_HIP_CLASS_SHARES5=$s
  timeout -k 10 200 python bench.py --atoms $ATOMS --no-cpu-baseline --primary-only --steps 300 --warmup 30 --condition-steps 500 > $OUT/cls.log 2>/dev/null || exit 1
  echo "F $ATOMS $s $(grep -o '"ms_per_step": [0-9.]*' $OUT/cls.log) $(grep -o '"k_calc_nb": [0-9.]*' $OUT/cls.log)" | tee -a $OUT/cls5.txt
done; done
