#!/bin/bash
# Round 3's verification session (like gpu_round2.sh): parity tests, the default bench line, kernel stats of the timed loop, sizes,
# kernel flavours, the domain step on one GPU.  usage: gpu_round3.sh TAG
set -u
OUT=gpurun_out
TAG=${1:-r3}
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
step() {   # name, timeout, command...
    local name=$1 tmo=$2; shift 2
    echo "=== $name ($(date +%T))"
    timeout -k 10 $tmo "$@" > $OUT/$name.log 2>&1
    local rc=$?
    echo "--- $name exit $rc"
    tail -n 6 $OUT/$name.log
    if [ $rc -ne 0 ]; then echo "!!! $name failed: stopping the session"; exit $rc; fi
    return 0
}
step ${TAG}_pytest_gpu 1000 python -m pytest tests -m gpu -x -q
step ${TAG}_bench 500 python bench.py
step ${TAG}_sizes 900 bash tools/gpu_sizes.sh
step ${TAG}_flavours 600 python tools/flavour_probe.py 300
for m in two merged; do
  step ${TAG}_ddprobe_$m 300 python tools/dd_single_gpu_probe.py 96k xyz rccl $m
done
step ${TAG}_ddprobe_1m_merged 300 python tools/dd_single_gpu_probe.py 768k xyz rccl merged
echo "=== done"
