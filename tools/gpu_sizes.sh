#!/bin/bash
# Fused force step against system size (fixed cost vs per-pair cost), BASELINE configs[1] (24k, here with Ewald), [2] (96k), [4] (768k)
OUT=gpurun_out; mkdir -p $OUT
for a in 24k 96k 768k; do
  timeout -k 10 400 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --atoms $a > $OUT/sz_$a.log 2>&1 || exit 1
  echo "$a $(grep -o '"ms_per_step": [0-9.]*' $OUT/sz_$a.log) $(grep -o '"value": [0-9.]*' $OUT/sz_$a.log) $(grep -o '"k_calc_nb": [0-9.]*' $OUT/sz_$a.log) $(grep -o '"cluster_pairs": [0-9]*' $OUT/sz_$a.log) $(grep -o '"ms_per_gpu_resident_md_step": [0-9.]*' $OUT/sz_$a.log)"
done
