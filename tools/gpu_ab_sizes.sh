#!/bin/bash
# force and energy step by box size: tree (age-class shares interpolated by range length) against variants/$1.so
OUT=gpurun_out; mkdir -p $OUT; V=${1:-head}; : > $OUT/r4af.txt
if [ -z "${CFGS:-}" ]; then cfgs=("24k rf" "96k ewald" "192k ewald" "768k ewald" "1m ewald"); else IFS=";" read -ra cfgs <<< "$CFGS"; fi
for cfg in "${cfgs[@]}"; do set -- $cfg
for rep in 1 2; do for lib in "" variants/$V.so; do
  [ -n "$lib" ] && export NBNXM_HIP_LIB=$lib || unset NBNXM_HIP_LIB
  r=$(timeout -k 10 300 python bench.py --atoms $1 --elec $2 --no-cpu-baseline --steps 100 --condition-steps 500 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('force %.4f ms energy %.4f dhdl %.4f md %.4f' % (r['ms_per_step'], r.get('ms_per_energy_step'), r.get('ms_per_dhdl_step_11_foreign_lambdas'), r.get('ms_per_gpu_resident_md_step')))") || exit 1
  echo "$1 $2 ${lib:-tree}: $r" | tee -a $OUT/r4af.txt
done; done; done
