#!/bin/bash
# round 4: force step of several builds, alternating, at several boxes.  usage: gpu_r4ab.sh TAG "lib1 lib2 .." ["96k ewald;24k rf"] [reps]   (lib "tree" = the working tree's library)
OUT=gpurun_out; mkdir -p $OUT; TAG=$1; LIBS=$2; CFGS=${3:-96k ewald;24k rf}; REPS=${4:-3}
: > $OUT/${TAG}.txt
IFS=';' read -ra CFGA <<< "$CFGS"
for cfg in "${CFGA[@]}"; do set -- $cfg
for rep in $(seq $REPS); do for lib in $LIBS; do
  [ "$lib" != "tree" ] && export NBNXM_HIP_LIB=variants/$lib.so || unset NBNXM_HIP_LIB
  r=$(timeout -k 10 200 python bench.py --atoms $1 --elec $2 --primary-only --no-cpu-baseline ${BENCH_EXTRA:-} 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms kernel %.2f us' % (r['ms_per_step'], r['kernel_us']['k_calc_nb']))") || exit 1
  echo "$1 $2 $lib: $r" | tee -a $OUT/${TAG}.txt
done; done; done
