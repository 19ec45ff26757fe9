#!/bin/bash
# force (or energy: STEP=energy) step against the cost-model weights slot,group,entry of the work partition (NBNXM_HIP_WORK_WEIGHTS; diagnostics)
OUT=gpurun_out; mkdir -p $OUT
export NBNXM_HIP_DIAGNOSTICS=1
EXTRA=""; [ "${STEP:-force}" = energy ] && EXTRA="--timed-step energy"
for rep in 1 2; do
for s in "$@"; do
  if [ "$s" = none ]; then unset NBNXM_HIP_WORK_WEIGHTS; else export NBNXM_HIP_WORK_WEIGHTS=$s; fi
  timeout -k 10 200 python bench.py --atoms ${ATOMS:-1m} --no-cpu-baseline --primary-only $EXTRA --steps 300 --warmup 30 --condition-steps 500 > $OUT/cls.log 2>/dev/null || exit 1
  echo "${ATOMS:-1m} ${STEP:-force} $s $(grep -o '"ms_per_step": [0-9.]*' $OUT/cls.log)" | tee -a $OUT/weights_sweep.txt
done; done
