#!/bin/bash
# round 4: what the slow waves of a launch have in common (cost-model fit with the executed-block feature), 24k RF and 96k Ewald; timeline budgets on the pruned list
OUT=gpurun_out; mkdir -p $OUT; L=${1:-variants/timeline.so}
for cfg in "24k rf" "96k ewald"; do set -- $cfg
  CAL_BOX=$1 CAL_ELEC=$2 NBNXM_HIP_LIB=$L timeout -k 10 300 python tools/calibrate_weights.py --executed 2>&1 | grep -v amdgpu.ids > $OUT/r4c_calibrate_$1_$2.txt || exit 1
  cat $OUT/r4c_calibrate_$1_$2.txt
  NBNXM_HIP_LIB=$L timeout -k 10 300 python tools/timeline_budget.py $cfg 2>&1 | grep -v amdgpu.ids > $OUT/r4c_budget_$1_$2.txt || exit 1
  cat $OUT/r4c_budget_$1_$2.txt
done
