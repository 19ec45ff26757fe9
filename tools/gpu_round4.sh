#!/bin/bash
# Round 4's verification session: parity tests, the default bench line, kernel stats + counters of the timed loop (force and energy
# flavours), sizes, the wave timelines behind DESIGN §4.1's budget, the domain step on one GPU.  usage: gpu_round4.sh COMMIT
set -u
OUT=gpurun_out
C=${1:-unknown}
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
step() {   # name, timeout, command...
    local name=$1 tmo=$2; shift 2
    echo "=== $name ($(date +%T))"
    timeout -k 10 $tmo "$@" > $OUT/$name.log 2>&1
    local rc=$?
    echo "--- $name exit $rc"
    tail -n 4 $OUT/$name.log
    if [ $rc -ne 0 ]; then echo "!!! $name failed: stopping the session"; exit $rc; fi
    return 0
}
step r4_pytest_gpu 1000 python -m pytest tests -m gpu -x -q
step r4_bench 500 python bench.py
step r4_pmc 1100 bash tools/gpu_r4_pmc.sh $C
step r4_sizes 900 bash tools/gpu_sizes.sh
for cfg in "24k rf force" "96k ewald force" "96k ewald energy" "96k ewald dhdl"; do set -- $cfg
  echo "=== timeline $1 $2 $3"
  TIMELINE_PROLOGUE=1 TIMELINE_STEP=$3 NBNXM_HIP_LIB=variants/timeline.so timeout -k 10 200 python tools/timeline_budget.py $1 $2 2>&1 | grep -v amdgpu.ids > $OUT/r4_budget_$1_$2_$3.txt || exit 1
  sed -n 1,14p $OUT/r4_budget_$1_$2_$3.txt
done
step r4_dd_probe 900 bash tools/gpu_r4n.sh
echo "=== done"
