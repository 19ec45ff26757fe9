#!/bin/bash
# energy / dH/dlambda step A/B of the working tree against variants (no tests)
OUT=gpurun_out; mkdir -p $OUT; V=${1:-c4}; TAG=${2:-r4z}
: > $OUT/${TAG}_ab.txt
if [ -z "${CFGS:-}" ]; then cfgs=("96k ewald" "24k rf"); else IFS=";" read -ra cfgs <<< "$CFGS"; fi
for cfg in "${cfgs[@]}"; do set -- $cfg
for rep in 1 2 3; do for lib in "" $(for v in $V; do echo variants/$v.so; done); do
  [ -n "$lib" ] && export NBNXM_HIP_LIB=$lib || unset NBNXM_HIP_LIB
  r=$(timeout -k 10 300 python bench.py --atoms $1 --elec $2 --no-cpu-baseline --steps 100 --condition-steps 500 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('force %.4f ms energy %.4f dhdl %.4f md %.4f' % (r['ms_per_step'], r.get('ms_per_energy_step'), r.get('ms_per_dhdl_step_11_foreign_lambdas'), r.get('ms_per_gpu_resident_md_step')))") || exit 1
  echo "$1 $2 ${lib:-tree}: $r" | tee -a $OUT/${TAG}_ab.txt
done; done; done
