#!/bin/bash
# A/B of the search step's GPU side: the working tree's library against variants/$1.so, alternating (bench.py's ms_per_search_step_gpu_side)
OUT=gpurun_out; mkdir -p $OUT; V=$1; : > $OUT/ab_search_$V.txt
for rep in 1 2 3; do for lib in "" variants/$V.so; do
  [ -n "$lib" ] && export NBNXM_HIP_LIB=$lib || unset NBNXM_HIP_LIB
  r=$(timeout -k 10 200 python bench.py --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms per step, search step %.3f ms' % (r['ms_per_step'], r['ms_per_search_step_gpu_side']))")
  echo "${lib:-tree}: $r" | tee -a $OUT/ab_search_$V.txt
done; done
