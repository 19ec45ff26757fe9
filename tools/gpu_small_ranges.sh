#!/bin/bash
# small boxes: force step against the number of work ranges (fewer, longer ranges pay the per-wave prologue less often)
OUT=gpurun_out; mkdir -p $OUT
for atoms in 24k 3k; do for n in 0 1024 2048 3072 4096; do
  NBNXM_HIP_NUM_WORK_RANGES=$n timeout -k 10 200 python bench.py --no-cpu-baseline --primary-only --steps 400 --warmup 40 --atoms $atoms > $OUT/sr.log 2>/dev/null || { echo FAILED; exit 1; }
  echo "$atoms ranges=$n $(grep -o '"ms_per_step": [0-9.]*' $OUT/sr.log)" | tee -a $OUT/small_ranges.txt
done; done
