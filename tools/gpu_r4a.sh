#!/bin/bash
# round 4, first session: where the launch goes (per-wave timeline budget at 24k RF, 24k Ewald, 96k, 1m) + the sizes line of the shipped library
OUT=gpurun_out; mkdir -p $OUT
for cfg in "24k rf" "24k ewald" "96k ewald" "96k rf"; do
  NBNXM_HIP_LIB=variants/timeline.so timeout -k 10 300 python tools/timeline_budget.py $cfg 2>&1 | grep -v amdgpu.ids | tee $OUT/r4a_budget_${cfg// /_}.txt || exit 1
done
for cfg in "24k rf" "96k ewald"; do
  set -- $cfg
  timeout -k 10 300 python bench.py --atoms $1 --elec $2 --primary-only --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/r4a_bench_$1_$2.json || exit 1
  python -c "import json,sys; r=json.load(open('$OUT/r4a_bench_$1_$2.json')); print('$cfg', r['ms_per_step'], r['kernel_us'])"
done
