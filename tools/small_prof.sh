cd /root/repo
for merged in 1 0; do
  echo "== NBNXM_HIP_FEP_MERGED=$merged"
  NBNXM_HIP_FEP_MERGED=$merged timeout -k 10 100 python bench.py --atoms 3k --no-cpu-baseline --primary-only --steps 1000 --warmup 100 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['fep_pairs'])"
done
export TMPDIR=/tmp
rm -rf gpurun_out/small_prof
NBNXM_HIP_FEP_MERGED=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/small_prof -- python3 bench.py --atoms 3k --no-cpu-baseline --primary-only --steps 300 --warmup 30 > gpurun_out/small_prof.log 2>&1
f=$(find gpurun_out/small_prof -name "*kernel_stats.csv" | head -1); cut -c1-90 $f | head -6; python3 -c "
import csv,sys
for r in list(csv.DictReader(open('$f')))[:5]: print(r['Name'][:40], r['Calls'], r['AverageNs'])"
