#!/bin/bash
# Plain cluster-pair kernel time against system size (fixed cost vs per-pair cost)
OUT=gpurun_out; mkdir -p $OUT
export NBNXM_HIP_FEP_CONCURRENT=0
for a in 24k 96k 768k; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --mode split --steps 50 --warmup 10 --atoms $a > $OUT/sz_$a.log 2>&1 || exit 1
  echo "$a $(grep -o '"k_calc_nb": [0-9.]*' $OUT/sz_$a.log) $(grep -o '"cluster_pairs": [0-9]*' $OUT/sz_$a.log) $(grep -o '"cj_slots": [0-9]*' $OUT/sz_$a.log) $(grep -o '"nsci": [0-9]*' $OUT/sz_$a.log)"
done
