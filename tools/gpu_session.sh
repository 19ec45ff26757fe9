#!/bin/bash
# One GPU-box session: parity tests, bench (fused + split), rocprofv3 kernel trace.
# A step that was killed or timed out (exit 124/137/139) ends the session: no further GPU step is started.
set -u
OUT=gpurun_out
mkdir -p $OUT
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
step() {   # name, timeout, command...
    local name=$1 tmo=$2; shift 2
    echo "=== $name ($(date +%T))"
    timeout -k 10 $tmo "$@" > $OUT/$name.log 2>&1
    local rc=$?
    echo "--- $name exit $rc"
    tail -n 15 $OUT/$name.log
    if [ $rc -eq 124 ] || [ $rc -eq 137 ] || [ $rc -eq 139 ] || [ $rc -eq 134 ]; then
        echo "!!! $name was killed / crashed: stopping the session"
        exit $rc
    fi
    return 0
}
MODE=${1:-all}
if [ "$MODE" = "all" ] || [ "$MODE" = "test" ]; then
    step pytest_gpu 900 python -m pytest tests -m gpu -x -q
fi
if [ "$MODE" = "all" ] || [ "$MODE" = "bench" ]; then
    step bench_fused 400 python bench.py --steps 200 --warmup 20
    step bench_split 300 python bench.py --steps 200 --warmup 20 --mode split --no-cpu-baseline
    rm -rf $OUT/prof
    step rocprof_stats 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --primary-only
    find $OUT/prof -name "*stats*" | head
fi
echo "=== done"
