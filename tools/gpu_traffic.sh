#!/bin/bash
# HBM traffic of the dominant kernel from PMC counters (separate passes, as MI355X_MICROARCH.md §HBM prescribes):
# FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of a wide coalesced read stream.
set -u
OUT=gpurun_out; mkdir -p $OUT; cd "$(dirname "$0")/.."
export TMPDIR=/tmp
rm -rf $OUT/traffic_*
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/traffic_$c -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --primary-only ${BENCH_EXTRA:-} > $OUT/traffic_$c.log 2>&1
  rc=$?; echo "--- $c exit $rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ] || [ $rc -eq 134 ]; then exit $rc; fi
done
python3 - <<'PY'
import csv, glob, json, collections
res = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for fn in glob.glob('gpurun_out/traffic_%s/**/*counter_collection.csv' % c, recursive=True):
        vals = collections.defaultdict(list)
        for row in csv.DictReader(open(fn)):
            if row['Counter_Name'] == c:
                vals[row['Kernel_Name']].append(float(row['Counter_Value']))
        for k, v in vals.items():
            # the force-only flavour: nbnxmKernel<ELEC, TWIN, VDW, ENERGY = false, FUSED>
            import re
            import os
            if re.search(r'nbnxmKernel<\d+, (false|true), \d+, %s, ' % os.environ.get('PMC_FLAVOUR_ENERGY', 'false'), k):
                res[c] = sum(v) / len(v)
                res['kernel'] = k[:70]
                res['n_' + c] = len(v)
if 'FETCH_SIZE' in res and 'WRITE_SIZE' in res:
    res['fetch_bytes_raw'] = res['FETCH_SIZE'] * 1024
    res['write_bytes'] = res['WRITE_SIZE'] * 1024
    # guide: FETCH_SIZE = 1/2 of the bytes of wide coalesced streams on gfx950 -> upper bound with x2
    res['hbm_bytes_per_launch_lower'] = res['fetch_bytes_raw'] + res['write_bytes']
    res['hbm_bytes_per_launch_corrected'] = 2 * res['fetch_bytes_raw'] + res['write_bytes']
print(json.dumps(res))
open('gpurun_out/traffic.json', 'w').write(json.dumps(res))
PY
