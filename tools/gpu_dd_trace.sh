#!/bin/bash
# kernel-trace timelines of the one-GPU domain-decomposition probe: two localities against merged localities
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
for m in two merged; do
  rm -rf $OUT/ddt_$m
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/ddt_$m -- python3 tools/dd_single_gpu_probe.py 96k xyz ${TRANSPORT:-rccl} $m > $OUT/ddt_$m.log 2>&1 || { tail -5 $OUT/ddt_$m.log; exit 1; }
  grep '^{' $OUT/ddt_$m.log | tail -n 1
  python3 tools/dd_timeline.py $OUT/ddt_$m 50 | tee $OUT/ddt_${m}_timeline.txt
  find $OUT/ddt_$m -name "*.csv" -size +2M -delete
done
