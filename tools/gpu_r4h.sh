#!/bin/bash
# round 4: sweep of the cost-model weights on short lists (NBNXM_HIP_WORK_WEIGHTS=slot,group,entry)
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/r4h_sweep.txt
for cfg in "24k rf" "12k ewald" "48k ewald"; do set -- $cfg
for rep in 1 2; do for w in "4,16,128" "0,48,200" "0,48,260" "0,64,260" "0,40,260" "4,48,200" "0,56,320" "0,64,400"; do
  r=$(NBNXM_HIP_WORK_WEIGHTS=$w timeout -k 10 200 python bench.py --atoms $1 --elec $2 --primary-only --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms kernel %.2f us' % (r['ms_per_step'], r['kernel_us']['k_calc_nb']))") || exit 1
  echo "$1 $2 weights $w: $r" | tee -a $OUT/r4h_sweep.txt
done; done; done
