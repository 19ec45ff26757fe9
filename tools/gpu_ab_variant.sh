#!/bin/bash
# A/B: the working tree's library against variants/$1.so, alternating.  usage: gpu_ab_variant.sh NAME [bench args]
OUT=gpurun_out; mkdir -p $OUT; V=$1; shift; : > $OUT/ab_$V.txt
ARGS=${@:---primary-only --no-cpu-baseline}
for rep in 1 2 3; do for lib in "" variants/$V.so; do
  [ -n "$lib" ] && export NBNXM_HIP_LIB=$lib || unset NBNXM_HIP_LIB
  r=$(timeout -k 10 200 python bench.py $ARGS 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms kernel %.2f us energy %s dhdl %s md %s' % (r['ms_per_step'], r['kernel_us']['k_calc_nb'], r.get('ms_per_energy_step'), r.get('ms_per_dhdl_step_11_foreign_lambdas'), r.get('ms_per_gpu_resident_md_step')))")
  echo "${lib:-tree}: $r" | tee -a $OUT/ab_$V.txt
done; done
