#!/bin/bash
# PMC passes over a short bench run (counters only with --kernel-trace, one group per pass).
set -u
OUT=gpurun_out
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
MODE=${1:-split}
shift || true
EXTRA="$@"
pass() {  # name, counters...
    local name=$1; shift
    echo "=== pmc $name: $@"
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --mode $MODE $EXTRA > $OUT/pmc_$name.log 2>&1
    local rc=$?
    echo "--- exit $rc"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ] || [ $rc -eq 134 ]; then exit $rc; fi
}
if [ ! -f $OUT/counters.txt ]; then rocprofv3 -L > $OUT/counters.txt 2>&1; fi
pass sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY
pass sq2 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA
pass sq3 SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_TRANS SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_INSTS_BRANCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
pass tcc1 TCC_EA0_ATOMIC_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
python3 - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc_*/')):
    files = glob.glob(d + '**/*counter_collection.csv', recursive=True)
    for fn in files:
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(fn)):
            acc[row['Kernel_Name'][:60]][row['Counter_Name']].append(float(row['Counter_Value']))
        for k, cs in acc.items():
            if 'nbnxm' not in k: continue
            print(d, k, {c: sum(v)/len(v) for c, v in cs.items()}, 'n=%d' % len(next(iter(cs.values()))))
PY
echo "=== done"
