#!/bin/bash
# 24k-atom box (BASELINE configs[1]): force step by the number of work ranges (default: one per wave slot, 5120)
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/ranges_24k.txt
for rep in 1 2; do for n in 0 1024 2048 3072 4096; do for elec in rf ewald; do
  r=$(NBNXM_HIP_NUM_WORK_RANGES=$n timeout -k 10 120 python bench.py --atoms 24k --elec $elec --primary-only --no-cpu-baseline --steps 2000 --warmup 100 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms kernel %.2f us' % (r['ms_per_step'], r['kernel_us']['k_calc_nb']))")
  echo "ranges $n $elec: $r" | tee -a $OUT/ranges_24k.txt
done; done; done
