#!/bin/bash
# energy / dH/dlambda step against the number of ranges of the energy flavours' partition (NBNXM_HIP_NUM_WORK_RANGES4; diagnostics): fewer
# ranges than wave slots leave slots free for the trailing work from the start of the kernel
OUT=gpurun_out; mkdir -p $OUT
export NBNXM_HIP_DIAGNOSTICS=1
if [ -z "${CFGS:-}" ]; then cfgs=("24k rf" "96k ewald" "1m ewald"); else IFS=";" read -ra cfgs <<< "$CFGS"; fi
for cfg in "${cfgs[@]}"; do set -- $cfg; box=$1; el=$2
for rep in 1 2; do for n in ${RANGES:-4096 3840 3584 3328 3072}; do
  export NBNXM_HIP_NUM_WORK_RANGES4=$n
  r=$(timeout -k 10 300 python bench.py --atoms $box --elec $el --no-cpu-baseline --steps 100 --condition-steps 500 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('force %.4f ms energy %.4f dhdl %.4f md %.4f' % (r['ms_per_step'], r.get('ms_per_energy_step'), r.get('ms_per_dhdl_step_11_foreign_lambdas'), r.get('ms_per_gpu_resident_md_step')))") || exit 1
  echo "$box $el ranges4 $n: $r" | tee -a $OUT/ranges4.txt
done; done; done
