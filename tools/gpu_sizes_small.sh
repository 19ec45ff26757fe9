#!/bin/bash
# small boxes: force / energy / MD step (the short-list rule of the work partition)
for atoms in 3k 12k 24k 48k; do for elec in ewald rf; do
  timeout -k 10 200 python bench.py --atoms $atoms --elec $elec --no-cpu-baseline --steps 1000 --warmup 100 2>/dev/null | grep '^{' | python -c "
import sys,json; r=json.loads(sys.stdin.read()); print('$atoms $elec force %.4f energy %.4f dhdl %.4f md %s' % (r['ms_per_step'], r['ms_per_energy_step'], r['ms_per_dhdl_step_11_foreign_lambdas'], r['ms_per_gpu_resident_md_step']))"
done; done
