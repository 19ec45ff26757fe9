#!/bin/bash
# round 4: cost-model fit of the current kernel at 24k RF / 48k / 96k Ewald, then a sweep of the weights (NBNXM_HIP_WORK_WEIGHTS=slot,group,entry)
OUT=gpurun_out; mkdir -p $OUT
for cfg in "24k rf" "96k ewald"; do set -- $cfg
  CAL_TAG=cur CAL_BOX=$1 CAL_ELEC=$2 NBNXM_HIP_LIB=variants/timeline.so timeout -k 10 300 python tools/calibrate_weights.py 2>&1 | grep -v amdgpu.ids | grep "^fit\|^weights\|^SIMDs\|residual" > $OUT/r4g_fit_$1.txt || exit 1
  cat $OUT/r4g_fit_$1.txt
done
: > $OUT/r4g_sweep.txt
for cfg in "24k rf" "96k ewald"; do set -- $cfg
for rep in 1 2; do for w in "4,16,128" "2,44,170" "0,40,116" "2,32,128" "0,48,200" "4,56,160" "0,30,90"; do
  r=$(NBNXM_HIP_WORK_WEIGHTS=$w timeout -k 10 200 python bench.py --atoms $1 --elec $2 --primary-only --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms kernel %.2f us' % (r['ms_per_step'], r['kernel_us']['k_calc_nb']))") || exit 1
  echo "$1 $2 weights $w: $r" | tee -a $OUT/r4g_sweep.txt
done; done; done
