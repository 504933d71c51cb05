#!/bin/bash
# sweep of launch-time knobs of the pipelined step on the driver's command:  scripts/gpu_sweep.sh <tag> "VAR=a VAR2=b" "VAR=c" ...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
out=$O/${TAG}_sweep.txt; : > $out
for cfg in "" "$@"; do
  v=$(env $cfg python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $BENCH_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%.1f k first %.1f k' % (d['value']/1e3, d['first_window']['value']/1e3))")
  echo "${cfg:-default}: $v" >> $out
done
cat $out
