#!/bin/bash
# Fused force step against system size and electrostatics: BASELINE configs[1] (24k, RF), [2] (96k, Ewald), [4]'s size (768k), and the 1M box
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/sizes.log
for cfg in "24k rf" "24k ewald" "96k rf" "96k ewald" "768k ewald" "1m ewald"; do
  set -- $cfg
  timeout -k 10 400 python bench.py --no-cpu-baseline --steps 200 --warmup 20 --atoms $1 --elec $2 > $OUT/sz_$1_$2.log 2>&1 || exit 1
  tail -n 1 $OUT/sz_$1_$2.log | python -c '
import json, sys
d = json.loads(sys.stdin.read())
g = lambda k: d.get(k) if d.get(k) is not None else float("nan")
print("%-5s %-6s cluster pairs %9d  force step %.4f ms  kernel %.1f us  %.3e pair-int/s  energy step %.4f  dhdl step %.4f  md step %.4f  frac %.3f" % (
    sys.argv[1], sys.argv[2], d["config"]["cluster_pairs"], d["ms_per_step"], d["kernel_us"]["k_calc_nb"], d["value"], g("ms_per_energy_step"),
    g("ms_per_dhdl_step_11_foreign_lambdas"), g("ms_per_gpu_resident_md_step"), d["roofline"]["frac"] if d.get("roofline") else float("nan")))' $1 $2 | tee -a $OUT/sizes.log
done
