#!/bin/bash
# round 4: energy / dH/dlambda steps with the perturbed pairs in a kernel of their own on a second stream BEHIND the cluster kernel
# (NBNXM_HIP_FEP_CONCURRENT=3, diagnostics) against the trailing workgroups (default) and the kernel ahead on the same stream
OUT=gpurun_out; mkdir -p $OUT; TAG=${1:-r4t}
: > $OUT/${TAG}.txt
for cfg in "96k ewald" "24k rf" "1m ewald"; do set -- $cfg
for rep in 1 2; do for mode in default behind ahead; do
  unset NBNXM_HIP_DIAGNOSTICS NBNXM_HIP_ENERGY_TAIL NBNXM_HIP_FEP_CONCURRENT
  [ $mode = behind ] && export NBNXM_HIP_DIAGNOSTICS=1 NBNXM_HIP_ENERGY_TAIL=1 NBNXM_HIP_FEP_CONCURRENT=3
  [ $mode = ahead ] && export NBNXM_HIP_DIAGNOSTICS=1 NBNXM_HIP_ENERGY_TAIL=1
  r=$(timeout -k 10 300 python bench.py --atoms $1 --elec $2 --no-cpu-baseline --steps 100 --condition-steps 500 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('force %.4f ms energy %.4f dhdl %.4f md %s' % (r['ms_per_step'], r.get('ms_per_energy_step'), r.get('ms_per_dhdl_step_11_foreign_lambdas'), r.get('ms_per_gpu_resident_md_step')))") || exit 1
  echo "$1 $2 $mode: $r" | tee -a $OUT/${TAG}.txt
done; done; done
