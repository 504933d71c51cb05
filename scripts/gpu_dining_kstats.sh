#!/bin/bash
# rocprofv3 kernel stats of bench.py --workload dining (run through gpurun) + the stage masks of one forward: gpurun_out/<tag>_dining_*.{csv,txt}
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_d; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_d -- python3 $R/bench.py --workload dining --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_dining_bench_prof.log 2>&1
f=$(find /tmp/prof_d -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $O/${TAG}_dining_kernel_stats.csv && head -8 $f | cut -c1-50,200-400
cd $R; timeout 600 python3 scripts/gpu_tree_phases.py dining > $O/${TAG}_dining_phases.txt 2>&1; cat $O/${TAG}_dining_phases.txt
