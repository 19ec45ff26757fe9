#!/bin/bash
# Step time of the 96k box against the number of atom types (LJ table in LDS), library A vs library B.
# usage: tools/gpu_types.sh libA.so libB.so
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/types.log
mkdir -p gpurun_out
: > $OUT
for extra in 0 26 44 60 96; do
  for lib in "$1" "$2"; do
    line=$(NBNXM_HIP_LIB=$lib timeout -k 10 200 python bench.py --extra-types $extra --steps 300 --warmup 30 --primary-only --no-cpu-baseline 2>/dev/null | tail -n 1)
    rc=$?
    if [ $rc -ne 0 ]; then echo "extra $extra $lib: exit $rc" | tee -a $OUT; exit $rc; fi
    python - "$extra" "$lib" "$line" <<'PY' | tee -a $OUT
import json, sys
d = json.loads(sys.argv[3])
print("types %3d  %-28s ms_per_step %.4f  kernel_us %.1f" % (d["config"]["atom_types"], sys.argv[2], d["ms_per_step"], d["kernel_us"]["k_calc_nb"]))
PY
  done
done
