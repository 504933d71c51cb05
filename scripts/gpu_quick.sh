#!/bin/bash
# quick GPU check: parity tests, short bench, kernel stats.  Usage: gpu_quick.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-190
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_q; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_q -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline > $O/$1_prof.log 2>&1
f=$(find /tmp/prof_q -name "*kernel_stats.csv" | head -1); cp $f $O/$1_kernel_stats.csv; head -7 $f | cut -c1-60,100-400 | awk -F, '{print $1, $(NF-6), $(NF-4), $(NF-2), $(NF-1)}'
