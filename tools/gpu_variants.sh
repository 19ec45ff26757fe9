#!/bin/bash
# Force-only step time of alternative builds of libnbnxm_hip.so (variants/*.so), each swapped in for the run
OUT=gpurun_out; mkdir -p $OUT
L=gromacs-fep-gpu_amd/lib/libnbnxm_hip.so
cp $L $OUT/lib_default.so.keep
for v in $L.orig variants/*.so; do
  if [ "$v" = "$L.orig" ]; then cp $OUT/lib_default.so.keep $L; name=default; else cp $v $L; name=$(basename $v); fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $OUT/var.log 2>&1 || { cp $OUT/lib_default.so.keep $L; exit 1; }
  echo "$name $(grep -o '"ms_per_step": [0-9.]*' $OUT/var.log) $(grep -o '"k_calc_nb": [0-9.]*' $OUT/var.log) $(grep -o '"ms_per_energy_step": [0-9.]*' $OUT/var.log)"
done
cp $OUT/lib_default.so.keep $L
