#!/bin/bash
# 24k RF force step against the class shares of the short-list partition (NBNXM_HIP_CLASS_SHARES_SHORT; diagnostics)
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/cls_short.txt
export NBNXM_HIP_DIAGNOSTICS=1
for rep in 1 2 3; do
for s in "$@"; do
  if [ "$s" = none ]; then unset NBNXM_HIP_CLASS_SHARES_SHORT; else export NBNXM_HIP_CLASS_SHARES_SHORT=$s; fi
  timeout -k 10 200 python bench.py --atoms ${ATOMS:-24k} --elec ${ELEC:-rf} --no-cpu-baseline --primary-only --steps 500 --warmup 50 --condition-steps 1000 > $OUT/cls.log 2>/dev/null || exit 1
  echo "${ATOMS:-24k} $s $(grep -o '"ms_per_step": [0-9.]*' $OUT/cls.log) $(grep -o '"k_calc_nb": [0-9.]*' $OUT/cls.log)" | tee -a $OUT/cls_short.txt
done; done
