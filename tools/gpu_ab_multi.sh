#!/bin/bash
# the working tree's library against several variants/NAME.so, alternating.  usage: gpu_ab_multi.sh "NAME1 NAME2 .." [bench args]
OUT=gpurun_out; mkdir -p $OUT; VARS=$1; shift; : > $OUT/ab_multi.txt
ARGS=${@:---primary-only --no-cpu-baseline}
for rep in 1 2 3; do for v in tree $VARS; do
  [ "$v" != tree ] && export NBNXM_HIP_LIB=variants/$v.so || unset NBNXM_HIP_LIB
  r=$(timeout -k 10 200 python bench.py $ARGS 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms kernel %.2f us' % (r['ms_per_step'], r['kernel_us']['k_calc_nb']))")
  echo "$v: $r" | tee -a $OUT/ab_multi.txt
done; done
