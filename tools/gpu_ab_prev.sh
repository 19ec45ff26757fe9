#!/bin/bash
# A/B: the working tree's library against variants/prev.so (tools/build_variant.sh prev on the previous commit's sources), alternating
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/ab_prev.txt
ARGS=${AB_ARGS:---primary-only --no-cpu-baseline}
for rep in 1 2 3; do for lib in "" variants/prev.so; do
  [ -n "$lib" ] && export NBNXM_HIP_LIB=$lib || unset NBNXM_HIP_LIB
  r=$(timeout -k 10 200 python bench.py $ARGS 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms kernel %.2f us energy %s dhdl %s md %s' % (r['ms_per_step'], r['kernel_us']['k_calc_nb'], r.get('ms_per_energy_step'), r.get('ms_per_dhdl_step_11_foreign_lambdas'), r.get('ms_per_gpu_resident_md_step')))")
  echo "${lib:-new}: $r" | tee -a $OUT/ab_prev.txt
done; done
