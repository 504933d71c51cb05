#!/bin/bash
# kernel duration statistics (rocprofv3 --kernel-trace --stats) of the driver's bench command for the default library and ab/ variants
#   scripts/gpu_kstats.sh <tag> [variant ...]   -> gpurun_out/<tag>_kstats_<variant>.csv (+ a summary on stdout)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for n in default "$@"; do
  rm -rf /tmp/ks_$n
  if [ "$n" = default ]; then unset SO101_HIP_LIB; else export SO101_HIP_LIB=$R/ab/lib_$n.so; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$n -- python3 $R/bench.py --steps 30 --warmup 3 --no-cpu-baseline $BENCH_ARGS > $O/${TAG}_kstats_$n.log 2>&1
  f=$(find /tmp/ks_$n -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] || { echo "== $n: no kernel_stats.csv (the profiler run failed)"; continue; }
  cp $f $O/${TAG}_kstats_$n.csv
  echo "== $n"; python3 - $f <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if r["Name"].startswith("k_"):
        print("  %-16s calls %6s  avg %8.1f us  min %7.1f  max %8.1f  total %8.1f ms" % (r["Name"].split("(")[0][:16], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
  grep -o '"value": [0-9.]*' $O/${TAG}_kstats_$n.log | head -1
done
