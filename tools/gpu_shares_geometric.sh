#!/bin/bash
# force-only step time for geometric shares of the age classes (share_k = r^k), default build and variants/*.so
OUT=gpurun_out; mkdir -p $OUT
L=gromacs-fep-gpu_amd/lib/libnbnxm_hip.so
cp $L $OUT/lib_default.so.keep
for v in $L.orig variants/*.so; do
  if [ "$v" = "$L.orig" ]; then cp $OUT/lib_default.so.keep $L; name=default; else cp $v $L; name=$(basename $v); fi
  for r in "$@"; do
    export NBNXM_HIP_CLASS_SHARES5=$(python3 -c "r=$r; print(','.join(str(int(10000*r**k)) for k in range(5)))")
    timeout -k 10 300 python bench.py --no-cpu-baseline --steps 200 --warmup 20 > $OUT/var.log 2>&1 || { cp $OUT/lib_default.so.keep $L; exit 1; }
    echo "$name r=$r $NBNXM_HIP_CLASS_SHARES5 $(grep -o '"ms_per_step": [0-9.]*' $OUT/var.log) $(grep -o '"k_calc_nb": [0-9.]*' $OUT/var.log)"
  done
done
cp $OUT/lib_default.so.keep $L
