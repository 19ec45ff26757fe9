#!/bin/bash
# the parity tests of the small boxes under the diagnostics switches that restore older routings (each must stay green)
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/diag_matrix.txt
export NBNXM_HIP_DIAGNOSTICS=1
K="split_path or fused_path or foreign_lambda or softcore or outputs or swapped or heavy or rolling or prune or config1"
for combo in "NBNXM_HIP_ENERGY_TAIL=0" "NBNXM_HIP_ENERGY_TAIL=1" "NBNXM_HIP_F_DOUBLE_BUFFER=0" "NBNXM_HIP_FEP_MERGED=0" "NBNXM_HIP_FEP_LIST_MERGED=0" "NBNXM_HIP_PRUNE_MERGED=0" "NBNXM_HIP_FEP_CONCURRENT=3 NBNXM_HIP_ENERGY_TAIL=1" "NBNXM_HIP_WAVES_PER_BLOCK=2"; do
  r=$(env $combo timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$K" 2>&1 | tail -1)
  echo "$combo: $r" | tee -a $OUT/diag_matrix.txt
done
