#!/bin/bash
# usage: gpu_sweep_env24.sh VAR v1 v2 ...   as gpu_sweep_env.sh on the 24k-atom box
OUT=gpurun_out; mkdir -p $OUT
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 --atoms 24k > $OUT/env.log 2>&1 || { tail -5 $OUT/env.log; exit 1; }
  echo "$VAR=$v $(grep -o '"ms_per_step": [0-9.]*' $OUT/env.log) $(grep -o '"k_calc_nb": [0-9.]*' $OUT/env.log) $(grep -o '"k_calc_nb_fep": [0-9.]*' $OUT/env.log) $(grep -o '"ms_per_energy_step": [0-9.]*' $OUT/env.log)"
done
