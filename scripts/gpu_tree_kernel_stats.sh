#!/bin/bash
# rocprofv3 kernel stats of the general-tree engine's bench workloads (run through gpurun): gpurun_out/<tag>_<workload>_bench_kernel_stats.csv
#   usage: bash scripts/gpu_tree_kernel_stats.sh r04
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for w in aloha dining; do
  rm -rf /tmp/prof_t_$w; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_t_$w -- python3 $R/bench.py --workload $w --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_${w}_bench_prof.log 2>&1
  f=$(find /tmp/prof_t_$w -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f $O/${TAG}_${w}_bench_kernel_stats.csv && head -6 $f | cut -c1-60
done
