#!/bin/bash
# round 4: parity tests, then the working tree against variants/$1.so (default r3) at 96k Ewald and 24k RF, alternating; then the timeline budget
OUT=gpurun_out; mkdir -p $OUT; V=${1:-r3}; TAG=${2:-r4b}
if [ "${SKIP_TESTS:-0}" != "1" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_pytest.log 2>&1; rc=$?; tail -5 $OUT/${TAG}_pytest.log
[ $rc -ne 0 ] && { echo "tests failed"; exit $rc; }
fi
: > $OUT/${TAG}_ab.txt
for cfg in "96k ewald" "24k rf"; do set -- $cfg
for rep in 1 2 3; do for lib in "" variants/$V.so; do
  [ -n "$lib" ] && export NBNXM_HIP_LIB=$lib || unset NBNXM_HIP_LIB
  r=$(timeout -k 10 200 python bench.py --atoms $1 --elec $2 --primary-only --no-cpu-baseline 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%.4f ms kernel %.2f us energy %s dhdl %s' % (r['ms_per_step'], r['kernel_us']['k_calc_nb'], r.get('ms_per_energy_step'), r.get('ms_per_dhdl_step_11_foreign_lambdas')))") || exit 1
  echo "$1 $2 ${lib:-tree}: $r" | tee -a $OUT/${TAG}_ab.txt
done; done; done
unset NBNXM_HIP_LIB
for cfg in "24k rf" "96k ewald"; do
  NBNXM_HIP_LIB=variants/timeline.so timeout -k 10 300 python tools/timeline_budget.py $cfg 2>&1 | grep -v amdgpu.ids > $OUT/${TAG}_budget_${cfg// /_}.txt || exit 1
  head -12 $OUT/${TAG}_budget_${cfg// /_}.txt
done
