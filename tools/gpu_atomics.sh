#!/bin/bash
# What do the memory-side float atomics cost the cluster kernel?  (1) the micro-benchmark of access shapes and cache-policy bits,
# (2) timing-only builds of the library (tools/build_variant.sh NAME -DNBNXM_TIMING_NO_{J_ATOMIC,I_ATOMIC,J_INSTR}: forces are
# wrong by construction, only the step time means something), alternating with the default build, at 96k and 1M atoms.
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 120 tools/ubench/atomic_shapes > $OUT/atomic_shapes.txt 2>&1 || { echo "ubench failed"; tail -5 $OUT/atomic_shapes.txt; exit 1; }
cat $OUT/atomic_shapes.txt
L=gromacs-fep-gpu_amd/lib/libnbnxm_hip.so
bash tools/gpu_ab.sh $L variants/noj.so variants/noi.so variants/noij.so variants/noinstr.so | tee $OUT/atomics_ab_96k.txt || exit 1
ARGS="--no-cpu-baseline --primary-only --steps 100 --warmup 10 --atoms 1m"
for v in $L variants/noj.so variants/noi.so variants/noij.so variants/noinstr.so; do
  NBNXM_HIP_LIB=$v timeout -k 10 300 python bench.py $ARGS > $OUT/ab.log 2>/dev/null || { echo "$v FAILED"; tail -3 $OUT/ab.log; exit 1; }
  echo "1m $(basename $v) $(grep -o '"ms_per_step": [0-9.]*' $OUT/ab.log)" | tee -a $OUT/atomics_ab_1m.txt
done
