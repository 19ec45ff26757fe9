#!/bin/bash
# as gpu_variants.sh, for the 96k and the 768k box, with equal shares of the ranges (NBNXM_HIP_CLASS_SHARES5)
OUT=gpurun_out; mkdir -p $OUT
L=gromacs-fep-gpu_amd/lib/libnbnxm_hip.so
cp $L $OUT/lib_default.so.keep
export NBNXM_HIP_CLASS_SHARES5=1024,1024,1024,1024,1024
for v in $L.orig variants/*.so; do
  if [ "$v" = "$L.orig" ]; then cp $OUT/lib_default.so.keep $L; name=default; else cp $v $L; name=$(basename $v); fi
  for a in 96k 768k; do
    timeout -k 10 300 python bench.py --no-cpu-baseline --steps 100 --warmup 10 --atoms $a > $OUT/var.log 2>&1 || { cp $OUT/lib_default.so.keep $L; exit 1; }
    echo "$name $a $(grep -o '"ms_per_step": [0-9.]*' $OUT/var.log) $(grep -o '"k_calc_nb": [0-9.]*' $OUT/var.log)"
  done
done
cp $OUT/lib_default.so.keep $L
