#!/bin/bash
# Domain step with a self-neighbouring rank (real RCCL): the local launch whole against in two parts, several first-part fractions
cd "$(dirname "$0")/.."
for rep in 1 2; do
  for cfg in "1 0.65" "2 0.5" "2 0.65" "2 0.8"; do
    set -- $cfg
    HALO_GPU_LOCAL_PARTS=$1 HALO_GPU_LOCAL_PART_FRACTION=$2 timeout -k 10 200 python tools/dd_single_gpu_probe.py ${ATOMS:-96k} xyz 2>/dev/null | grep "^{" | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('parts $1 fraction $2: ms_per_step %.4f host %.4f' % (d['ms_per_step'], d['ms_host_enqueue_per_step']))"
  done
done
