#!/bin/bash
# timeline budgets of two instrumented builds on the same (pruned) lists
OUT=gpurun_out; mkdir -p $OUT
for cfg in "24k rf" "96k ewald"; do for L in timeline_r3 timeline; do
  echo "=== $L $cfg"
  NBNXM_HIP_LIB=variants/$L.so timeout -k 10 300 python tools/timeline_budget.py $cfg 2>&1 | grep -v amdgpu.ids > $OUT/r4e_budget_${L}_${cfg// /_}.txt || exit 1
  cat $OUT/r4e_budget_${L}_${cfg// /_}.txt
  cp $OUT/timeline_budget_${cfg// /_}.npy $OUT/r4e_${L}_${cfg// /_}.npy
done; done
