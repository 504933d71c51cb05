#!/bin/bash
# A/B of kernel variants on one box: bench.py (driver's command) for the default library and every ab/lib_<name>.so given.
#   scripts/gpu_ab.sh <tag> name1 name2 ...      -> gpurun_out/<tag>_ab.txt
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
out=$O/${TAG}_ab.txt; : > $out
run() {  # name libpath extra-args
  local n=$1 lib=$2; shift; shift
  for rep in 1 2; do
    if [ -n "$lib" ]; then SO101_HIP_LIB=$lib python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > /tmp/ab_$n.json 2>/tmp/ab_$n.err; else python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > /tmp/ab_$n.json 2>/tmp/ab_$n.err; fi
    python3 - "$n" /tmp/ab_$n.json "$@" >> $out <<'PY'
import json, sys
n, f = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{n:24s} {' '.join(sys.argv[3:]):28s} value {d['value']/1e3:8.1f} k  first {d.get('first_window', {}).get('value', 0)/1e3 if isinstance(d.get('first_window'), dict) else 0:8.1f} k  ms {d['ms_per_step']:.3f}")
except Exception as ex:
    print(n, 'FAILED', ex)
PY
  done
}
run default ""
for n in "$@"; do run $n $R/ab/lib_$n.so; done
if [ -n "$AB_BIG" ]; then
  run default "" --envs-per-gpu 32768 --steps 10
  for n in "$@"; do run $n $R/ab/lib_$n.so --envs-per-gpu 32768 --steps 10; done
fi
cat $out
