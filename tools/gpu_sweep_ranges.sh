#!/bin/bash
# Sweep of the number of work ranges (waves) of the cluster-pair kernel; plain kernel, FEP kernels serialized.
OUT=gpurun_out; mkdir -p $OUT
export NBNXM_HIP_FEP_CONCURRENT=0
for n in ${SWEEP:-1024 2048 3072 4096 5120 6144 8192 10240 15360 20480 40960}; do
  NBNXM_HIP_NUM_WORK_RANGES=$n timeout -k 10 200 python bench.py --no-cpu-baseline --mode ${MODE:-split} --steps 100 --warmup 10 ${EXTRA:-} > $OUT/sw_$n.log 2>&1 || exit 1
  echo "$n $(grep -o '"k_calc_nb": [0-9.]*' $OUT/sw_$n.log) $(grep -o '"ms_per_step": [0-9.]*' $OUT/sw_$n.log)"
done
