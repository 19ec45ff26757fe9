#!/bin/bash
# force step against the number of work ranges (0 = default, one per wave slot) by box size: where do fewer, longer ranges pay?
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/ranges_sizes.txt
for atoms in 12k 24k 48k 96k; do for n in 0 4096 3072; do for rep in 1 2; do
  r=$(NBNXM_HIP_NUM_WORK_RANGES=$n timeout -k 10 120 python bench.py --atoms $atoms --primary-only --no-cpu-baseline --steps 1500 --warmup 100 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms  cj slots %d' % (r['ms_per_step'], r['config']['cj_slots']))")
  echo "$atoms ranges $n: $r" | tee -a $OUT/ranges_sizes.txt
done; done; done
