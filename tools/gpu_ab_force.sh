#!/bin/bash
# force step of the tree against variants, alternating (primary figure only)
OUT=gpurun_out; mkdir -p $OUT; V=${1:-head}
if [ -z "${CFGS:-}" ]; then cfgs=("96k ewald" "1m ewald"); else IFS=";" read -ra cfgs <<< "$CFGS"; fi
for cfg in "${cfgs[@]}"; do set -- $cfg
for rep in 1 2; do for lib in "" $(for v in $V; do echo variants/$v.so; done); do
  [ -n "$lib" ] && export NBNXM_HIP_LIB=$lib || unset NBNXM_HIP_LIB
  timeout -k 10 200 python bench.py --atoms $1 --elec $2 --no-cpu-baseline --primary-only ${EXTRA:-} --steps 300 --warmup 30 --condition-steps 500 > $OUT/cls.log 2>/dev/null || exit 1
  echo "$1 $2 ${lib:-tree} $(grep -o '"ms_per_step": [0-9.]*' $OUT/cls.log)"
done; done; done
