#!/bin/bash
# round 4: one-sided transport: tests, the one-GPU probe against RCCL (alternating), and a kernel-trace timeline of a steady-state step of both
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_sided" 2>&1 | tail -2 || exit 1
bash tools/gpu_r4n.sh || exit 1
for t in rccl push; do
  rm -rf $OUT/ddt_$t
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/ddt_$t -- python3 tools/dd_single_gpu_probe.py 96k xyz $t merged > $OUT/ddt_$t.log 2>&1 || { tail -5 $OUT/ddt_$t.log; exit 1; }
  python3 tools/dd_timeline.py $OUT/ddt_$t 50 | tee $OUT/r4o_dd_timeline_$t.txt
  find $OUT/ddt_$t -name "*.csv" -size +2M -delete
done
