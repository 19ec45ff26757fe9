#!/bin/bash
# Force-only step time (primary figure only, two runs each) of alternative builds of libnbnxm_hip.so (variants/*.so)
OUT=gpurun_out; mkdir -p $OUT
L=gromacs-fep-gpu_amd/lib/libnbnxm_hip.so
cp $L $OUT/lib_default.so.keep
for v in $L.orig variants/*.so $L.orig; do
  if [ "$v" = "$L.orig" ]; then cp $OUT/lib_default.so.keep $L; name=default; else cp $v $L; name=$(basename $v); fi
  for rep in 1 2; do
    timeout -k 10 200 python bench.py --no-cpu-baseline --primary-only --steps 300 --warmup 30 > $OUT/var.log 2>/dev/null || { cp $OUT/lib_default.so.keep $L; exit 1; }
    echo "$name $(grep -o '"ms_per_step": [0-9.]*' $OUT/var.log) $(grep -o '"k_calc_nb": [0-9.]*' $OUT/var.log)"
  done
done
cp $OUT/lib_default.so.keep $L
