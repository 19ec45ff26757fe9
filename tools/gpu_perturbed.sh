#!/bin/bash
# Step time of the 96k box against the size of the perturbed region (3 atoms per perturbed molecule), fused and split mode.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/perturbed.log
mkdir -p gpurun_out
: > $OUT
for npert in ${PERT:-16 50 150 500}; do
  for mode in fused split; do
    line=$(timeout -k 10 200 python bench.py --perturbed-molecules $npert --mode $mode --steps 300 --warmup 30 --no-cpu-baseline 2>gpurun_out/perturbed_err.log | tail -n 1)
    rc=$?
    if [ $rc -ne 0 ]; then echo "npert $npert $mode: exit $rc" | tee -a $OUT; tail -n 5 gpurun_out/perturbed_err.log | tee -a $OUT; continue; fi
    python - "$mode" "$line" <<'PY' | tee -a $OUT
import json, sys
d = json.loads(sys.argv[2])
c = d["config"]
print("perturbed atoms %5d  %-5s fep_pairs %8d  ms_per_step %.4f  energy step %.4f  dhdl step %.4f" % (
    c["perturbed_atoms"], sys.argv[1], c["fep_pairs"], d["ms_per_step"], d.get("ms_per_energy_step", float("nan")),
    d.get("ms_per_dhdl_step_11_foreign_lambdas", float("nan"))))
PY
  done
done
