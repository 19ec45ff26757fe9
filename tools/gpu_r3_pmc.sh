#!/bin/bash
# round 3: kernel stats + PMC passes of the default bench command at the current commit (merged by tools/summarize_counters.py afterwards)
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
rm -rf $OUT/pmc_* $OUT/traffic_* $OUT/r3_stats
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r3_stats -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --primary-only > $OUT/r3_stats_bench.json 2> $OUT/r3_stats.err || { tail -5 $OUT/r3_stats.err; exit 1; }
f=$(find $OUT/r3_stats -name "*kernel_stats.csv" | head -1); cp $f $OUT/r3_kernel_stats_fused.csv; head -5 $f
find $OUT/r3_stats -name "*.csv" -size +2M -delete
bash tools/gpu_pmc.sh fused --primary-only > $OUT/r3_pmc.log 2>&1 || { tail -20 $OUT/r3_pmc.log; exit 1; }
bash tools/gpu_traffic.sh > $OUT/r3_traffic.log 2>&1 || { tail -20 $OUT/r3_traffic.log; exit 1; }
tail -3 $OUT/r3_traffic.log
python3 tools/summarize_counters.py $OUT ${1:-unknown} $OUT/counters_fused_force_kernel.json
find $OUT/pmc_* $OUT/traffic_* -name "*.csv" -size +1M -delete
