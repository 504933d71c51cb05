#!/bin/bash
# Instruction-cache and scalar-cache counters of the step's kernels (separate --pmc pass, kernel-trace only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_IFETCH" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pi_$n; timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pi_$n -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-prefetch > $O/icache_$n.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
t = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob('/tmp/pi_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        for kk in ('k_narrow', 'k_pipe_solve', 'k_pipe_begin'):
            if kk in k:
                t[(kk, r['Counter_Name'])] += float(r['Counter_Value']); n[(kk, r['Counter_Name'])] += 1
for k in sorted(t): print("%-14s %-30s %.4g per dispatch (%d dispatches)" % (k[0], k[1], t[k] / n[k], n[k]))
PY
