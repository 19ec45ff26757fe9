#!/bin/bash
# round 3, session h: the atom-pair list regrouped by cluster pair in the cluster kernel's tail -- parity, then split-mode step time
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $OUT/r3h_pytest.log 2>&1 || { tail -40 $OUT/r3h_pytest.log; exit 1; }
tail -3 $OUT/r3h_pytest.log
for rep in 1 2; do
  timeout -k 10 200 python bench.py --mode split 2>/dev/null | grep '^{' | tee -a $OUT/r3h_split.json
  NBNXM_HIP_FEP_LIST_MERGED=0 timeout -k 10 200 python bench.py --mode split 2>/dev/null | grep '^{' | tee -a $OUT/r3h_split_unmerged.json
done
timeout -k 10 200 python bench.py 2>/dev/null | grep '^{' | tee $OUT/r3h_fused.json
