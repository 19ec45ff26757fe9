#!/bin/bash
# per-wave timeline of the force kernel incl. its trailing workgroups (diagnostics build variants/timeline.so: tools/build_variant.sh timeline -DNBNXM_WAVE_TIMELINE)
OUT=gpurun_out; mkdir -p $OUT
for mode in fused split; do
  echo "== $mode"; NBNXM_HIP_LIB=variants/timeline.so timeout -k 10 300 python tools/timeline.py $mode 96k 2>&1 | grep -v amdgpu.ids | tee $OUT/timeline_tail_$mode.txt
done
