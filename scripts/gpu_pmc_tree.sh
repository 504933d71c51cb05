#!/bin/bash
# Instruction-cache / issue counters of the general-tree engine's step kernel (separate --pmc passes, kernel-trace only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pt_$n; timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pt_$n -- python3 $R/scripts/gpu_aloha_bench.py banana > $O/tree_pmc_$n.log 2>&1
done
python3 - <<'PY'
import csv, glob, collections
t = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob('/tmp/pt_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if 'k_tree_step' in k and int(r.get('Grid_Size', r.get('Grid_Size_X', '0')) or 0) >= 4096 * 64:
            t[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
for k in sorted(t): print("k_tree_step (4096 envs)  %-30s %.4g per dispatch (%d dispatches)" % (k, t[k] / n[k], n[k]))
PY
