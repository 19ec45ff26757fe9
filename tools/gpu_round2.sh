#!/bin/bash
# One GPU-box session of round 2: parity tests, the default bench line, rocprofv3 kernel trace of the timed loop.
# A step that was killed or timed out ends the session: no further GPU step is started.
set -u
OUT=gpurun_out
TAG=${1:-a}
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
step() {   # name, timeout, command...
    local name=$1 tmo=$2; shift 2
    echo "=== $name ($(date +%T))"
    timeout -k 10 $tmo "$@" > $OUT/$name.log 2>&1
    local rc=$?
    echo "--- $name exit $rc"
    tail -n 12 $OUT/$name.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ] || [ $rc -eq 134 ]; then
        echo "!!! $name was killed / crashed: stopping the session"
        exit $rc
    fi
    return 0
}
step ${TAG}_pytest_gpu 1000 python -m pytest tests -m gpu -x -q
step ${TAG}_bench 400 python bench.py
rm -rf $OUT/${TAG}_prof
step ${TAG}_rocprof 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --primary-only
find $OUT/${TAG}_prof -name "*kernel_stats*" | head -3
echo "=== done"
