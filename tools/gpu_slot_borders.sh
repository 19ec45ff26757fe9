#!/bin/bash
# range borders at j-cluster slots: NBNXM_HIP_SLOT_BORDERS 0 (never) / 1 (auto: short lists' force partition) / 2 (always), by box size
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/r4ak.txt
export NBNXM_HIP_DIAGNOSTICS=1
if [ -z "${CFGS:-}" ]; then cfgs=("3k ewald" "12k rf" "24k rf" "24k ewald" "48k ewald"); else IFS=";" read -ra cfgs <<< "$CFGS"; fi
for cfg in "${cfgs[@]}"; do set -- $cfg
for rep in 1 2; do for m in 0 1 2; do
  export NBNXM_HIP_SLOT_BORDERS=$m
  r=$(timeout -k 10 300 python bench.py --atoms $1 --elec $2 --no-cpu-baseline --steps 200 --condition-steps 500 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('force %.4f ms energy %.4f dhdl %.4f md %.4f' % (r['ms_per_step'], r.get('ms_per_energy_step'), r.get('ms_per_dhdl_step_11_foreign_lambdas'), r.get('ms_per_gpu_resident_md_step')))") || exit 1
  echo "$1 $2 slot borders $m: $r" | tee -a $OUT/r4ak.txt
done; done; done
