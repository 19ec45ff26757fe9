#!/bin/bash
# Parameter sweep of the bench (list granularity x waves per block); prints kernel time per variant.
set -u
OUT=gpurun_out; mkdir -p $OUT; cd "$(dirname "$0")/.."
MODE=${1:-split}
for w in 1 2 4; do for c in 2 4 8 16; do
  export NBNXM_HIP_WAVES_PER_BLOCK=$w
  timeout -k 10 120 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --mode $MODE --max-cjpacked-per-sci $c > $OUT/sweep_${MODE}_w${w}_c${c}.log 2>&1
  rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ] || [ $rc -eq 134 ]; then echo "killed rc=$rc"; exit $rc; fi
  python3 - <<PY
import json
try:
    d=json.loads(open("$OUT/sweep_${MODE}_w${w}_c${c}.log").read().strip().splitlines()[-1])
    print("mode $MODE waves/block $w cjp/sci $c: nsci %d k_nb %.1f us fep %.1f us ms/step %.4f" % (d["config"]["nsci"], d["kernel_us"]["k_calc_nb"], d["kernel_us"]["k_calc_nb_fep"], d["ms_per_step"]))
except Exception as e:
    print("w$w c$c failed", e)
PY
done; done
