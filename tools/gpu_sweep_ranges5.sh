#!/bin/bash
# force step against the number of ranges of ALL partitions (NBNXM_HIP_NUM_WORK_RANGES; diagnostics; only the force figure means something)
OUT=gpurun_out; mkdir -p $OUT
export NBNXM_HIP_DIAGNOSTICS=1
if [ -z "${CFGS:-}" ]; then cfgs=("24k rf"); else IFS=";" read -ra cfgs <<< "$CFGS"; fi
for cfg in "${cfgs[@]}"; do set -- $cfg; box=$1; el=$2
for rep in 1 2 3; do for n in ${RANGES:-0 4608 4352 4096 3840 3584}; do
  if [ $n = 0 ]; then unset NBNXM_HIP_NUM_WORK_RANGES; else export NBNXM_HIP_NUM_WORK_RANGES=$n; fi
  timeout -k 10 200 python bench.py --atoms $box --elec $el --no-cpu-baseline --primary-only --steps 500 --warmup 50 --condition-steps 1000 > $OUT/cls.log 2>/dev/null || exit 1
  echo "$box $el ranges $n $(grep -o '"ms_per_step": [0-9.]*' $OUT/cls.log)" | tee -a $OUT/ranges5.txt
done; done; done
