#!/bin/bash
# force step: 4-wave workgroups (tree) against 2-wave workgroups with a 1,024-entry force table (variants/f1024.so), alternating
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/r4ae.txt
for cfg in "96k ewald" "24k ewald"; do set -- $cfg
for rep in 1 2 3; do for mode in tree f1024w2 f1024w4; do
  unset NBNXM_HIP_LIB NBNXM_HIP_DIAGNOSTICS NBNXM_HIP_WAVES_PER_BLOCK
  [ $mode = f1024w2 ] && export NBNXM_HIP_LIB=variants/f1024.so NBNXM_HIP_DIAGNOSTICS=1 NBNXM_HIP_WAVES_PER_BLOCK=2
  [ $mode = f1024w4 ] && export NBNXM_HIP_LIB=variants/f1024.so
  r=$(timeout -k 10 300 python bench.py --atoms $1 --elec $2 --no-cpu-baseline --primary-only --steps 200 --condition-steps 500 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('force %.4f ms kernel %.2f us' % (r['ms_per_step'], r['kernel_us']['k_calc_nb']))") || exit 1
  echo "$1 $2 $mode: $r" | tee -a $OUT/r4ae.txt
done; done; done
