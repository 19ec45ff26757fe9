#!/bin/bash
# round 4: kernel stats + PMC passes of the default bench command at the current commit, for the force-only kernel of the timed loop and
# (bench.py --timed-step energy) for the energy flavour; merged by tools/summarize_counters.py.  usage: gpu_r4_pmc.sh COMMIT
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
C=${1:-unknown}
for flav in force energy; do
  rm -rf $OUT/pmc_* $OUT/traffic_* $OUT/r4_stats_$flav
  EXTRA=""; [ $flav = energy ] && EXTRA="--timed-step energy"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/r4_stats_$flav -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --primary-only $EXTRA > $OUT/r4_stats_${flav}_bench.json 2> $OUT/r4_stats_$flav.err || { tail -5 $OUT/r4_stats_$flav.err; exit 1; }
  f=$(find $OUT/r4_stats_$flav -name "*kernel_stats.csv" | head -1); cp $f $OUT/r4_kernel_stats_fused_$flav.csv; head -4 $f
  find $OUT/r4_stats_$flav -name "*.csv" -size +2M -delete
  bash tools/gpu_pmc.sh fused --primary-only $EXTRA > $OUT/r4_pmc_$flav.log 2>&1 || { tail -20 $OUT/r4_pmc_$flav.log; exit 1; }
  if [ $flav = energy ]; then export PMC_FLAVOUR_ENERGY=true BENCH_EXTRA="--timed-step energy"; else unset PMC_FLAVOUR_ENERGY BENCH_EXTRA; fi
  bash tools/gpu_traffic.sh > $OUT/r4_traffic_$flav.log 2>&1 || { tail -20 $OUT/r4_traffic_$flav.log; exit 1; }
  tail -3 $OUT/r4_traffic_$flav.log
  python3 tools/summarize_counters.py $OUT $C $OUT/counters_fused_${flav}_kernel.json $flav
  find $OUT/pmc_* $OUT/traffic_* -name "*.csv" -size +1M -delete
done
