#!/bin/bash
# round 4: lambda windows batched into one object where the launch's fixed cost dominates (24k RF, configs[1]/[3]) and at 96k
OUT=gpurun_out; mkdir -p $OUT
for cfg in "24k rf" "96k ewald" "3k ewald"; do set -- $cfg
  timeout -k 10 400 python bench.py --atoms $1 --elec $2 --no-cpu-baseline 2>/dev/null | grep '^{' > $OUT/r4l_bench_$1_$2.json || exit 1
  python -c "
import json; r=json.load(open('$OUT/r4l_bench_$1_$2.json')); b=r['lambda_windows_batched']
print('$cfg: one window %.4f ms (energy %.4f) | 11 windows in one object %.4f ms = %.4f per window, x%.2f; energy step %.4f ms, x%.2f' % (r['ms_per_step'], r['ms_per_energy_step'], b['ms_per_step_all_windows'], b['ms_per_step_all_windows']/11, b['speedup_over_one_window_at_a_time'], b['ms_per_energy_step_all_windows'], b['energy_step_speedup_over_one_window_at_a_time']))
print('   cold %.4f  conditioned %.4f' % (r['ms_per_step_cold'], r['ms_per_step']))"
done
