#!/bin/bash
# round 4: the domain step on one GPU (96k home + 46k halo atoms, self links along x, y, z): RCCL merged against the one-sided transport, alternating
OUT=gpurun_out; mkdir -p $OUT; : > $OUT/r4n_dd_probe.txt
for rep in 1 2 3; do for t in rccl push; do
  r=$(timeout -k 10 300 python tools/dd_single_gpu_probe.py 96k xyz $t merged 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('%s: %.4f ms per step, host %.4f ms (empty queue %.4f)' % (r['transport'], r['ms_per_step'], r['ms_host_enqueue_per_step'], r['ms_host_enqueue_per_step_empty_queue']))") || exit 1
  echo "$r" | tee -a $OUT/r4n_dd_probe.txt
done; done
for t in rccl push; do
  r=$(timeout -k 10 300 python tools/dd_single_gpu_probe.py 768k xyz $t merged 2>/dev/null | grep '^{' | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('768k %s: %.4f ms per step, host %.4f ms' % (r['transport'], r['ms_per_step'], r['ms_host_enqueue_per_step']))") || exit 1
  echo "$r" | tee -a $OUT/r4n_dd_probe.txt
done
