#!/bin/bash
# Kernel trace of the domain-decomposition step with a self-neighbouring rank (real RCCL send / receive groups, one GPU) and of the energy steps
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
rm -rf $OUT/dd_prof $OUT/en_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/dd_prof -- python3 tools/dd_single_gpu_probe.py 96k xyz > $OUT/dd_prof.log 2>&1
echo "dd rc $?"; grep '^{' $OUT/dd_prof.log | tail -n 1
f=$(find $OUT/dd_prof -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv
for r in list(csv.DictReader(open('$f')))[:14]: print('%-70s %6s %10.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])))"
cp $f $OUT/dd_kernel_stats.csv
FLAVOURS=ewald/cut timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/en_prof -- python3 tools/flavour_probe.py 200 > $OUT/en_prof.log 2>&1
echo "energy rc $?"; grep "force step" $OUT/en_prof.log
f=$(find $OUT/en_prof -name "*kernel_stats.csv" | head -1)
python3 -c "
import csv
for r in list(csv.DictReader(open('$f')))[:8]: print('%-70s %6s %10.1f' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])))"
cp $f $OUT/energy_kernel_stats.csv
