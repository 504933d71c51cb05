#!/bin/bash
# A/B of general-tree engine variants on one box: bench.py --workload dining / aloha for the default library and every ab/lib_<name>.so given.
#   scripts/gpu_ab_tree.sh <tag> name1 name2 ...      -> gpurun_out/<tag>_ab_tree.txt      (AB_WORKLOADS="dining aloha" by default)
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
out=$O/${TAG}_ab_tree.txt; : > $out
for w in ${AB_WORKLOADS:-dining aloha}; do
  for n in default "$@"; do
    for rep in 1 2; do
      if [ "$n" = default ]; then lib=""; else lib=$R/ab/lib_$n.so; fi
      SO101_HIP_LIB=$lib python3 bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > /tmp/abt.json 2>/tmp/abt.err
      python3 - "$n" "$w" >> $out <<'PY'
import json, sys
try:
    d = json.loads(open('/tmp/abt.json').read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:8s} {sys.argv[1]:24s} value {d['value']/1e3:8.1f} k  ms {d['ms_per_step']:.3f}")
except Exception as ex:
    print(sys.argv[2], sys.argv[1], 'FAILED', ex, open('/tmp/abt.err').read()[-300:])
PY
    done
  done
done
cat $out
