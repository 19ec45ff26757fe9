#!/bin/bash
# usage: gpu_sweep_env.sh VAR v1 v2 ...   force-only step time of bench.py for each value of an environment variable
OUT=gpurun_out; mkdir -p $OUT
VAR=$1; shift
for v in "$@"; do
  export $VAR=$v
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $OUT/env.log 2>&1 || { tail -5 $OUT/env.log; exit 1; }
  echo "$VAR=$v $(grep -o '"ms_per_step": [0-9.]*' $OUT/env.log) $(grep -o '"k_calc_nb": [0-9.]*' $OUT/env.log) $(grep -o '"ms_per_energy_step": [0-9.]*' $OUT/env.log) $(grep -o '"ms_per_gpu_resident_md_step": [0-9.]*' $OUT/env.log)"
done
