#!/bin/bash
# sustained runs (episode ends and their resets inside the timed region) of the default library and ab/ variants:  scripts/gpu_sustained_ab.sh <tag> variant...
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O; cd $R
out=$O/${TAG}_sustained.txt; : > $out
for n in default "$@"; do
  if [ "$n" = default ]; then unset SO101_HIP_LIB; else export SO101_HIP_LIB=$R/ab/lib_$n.so; fi
  for steps in 500 1500; do
    python3 bench.py --steps $steps --warmup 10 --repeats 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); s = d['sustained']
print('%-10s steps %4d  host %6.1f k  device %6.1f k  per100 %s  events %s' % ('$n', $steps, d['value'] / 1e3, s['device_env_steps_per_s'] / 1e3, [round(x / 1e3) for x in s['per_100_steps_env_steps_per_s']], {k: round(v, 7) for k, v in d['events_per_env_step'].items() if v}))" >> $out
  done
done
cat $out
