#!/bin/bash
# round 4: energy / dH/dl steps of the working tree against variants/$1.so, alternating; 96k Ewald, 24k RF, 1m Ewald (energy only)
OUT=gpurun_out; mkdir -p $OUT; V=${1:-c2}; TAG=${2:-r4j}
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest.log 2>&1; rc=$?; tail -3 $OUT/${TAG}_pytest.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" $OUT/${TAG}_pytest.log | head -20; exit $rc; }
fi
: > $OUT/${TAG}_ab.txt
for cfg in "96k ewald" "24k rf" "1m ewald"; do set -- $cfg
for rep in 1 2; do for lib in "" variants/$V.so; do
  [ -n "$lib" ] && export NBNXM_HIP_LIB=$lib || unset NBNXM_HIP_LIB
  r=$(timeout -k 10 300 python bench.py --atoms $1 --elec $2 --no-cpu-baseline --steps 100 --condition-steps 500 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('force %.4f ms energy %s dhdl %s md %s' % (r['ms_per_step'], r.get('ms_per_energy_step'), r.get('ms_per_dhdl_step_11_foreign_lambdas'), r.get('ms_per_gpu_resident_md_step')))") || exit 1
  echo "$1 $2 ${lib:-tree}: $r" | tee -a $OUT/${TAG}_ab.txt
done; done; done
