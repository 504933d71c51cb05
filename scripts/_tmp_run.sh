R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for P in 2 1; do
for c in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQC_TC_STALL SQC_ICACHE_BUSY_CYCLES SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_INSTS_BRANCH"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pi_$n; PIPELINE=$P timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pi_$n -- python3 $R/scripts/gpu_pmc_step.py > /dev/null 2>&1
done
echo "== pipeline $P"
python3 - <<'PY'
import csv, glob, collections
t = collections.defaultdict(float); n = collections.defaultdict(int)
for f in glob.glob('/tmp/pi_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        for kk in ('k_narrow', 'k_pipe_solve', 'k_chain'):
            if kk in k:
                t[(kk, r['Counter_Name'])] += float(r['Counter_Value']); n[(kk, r['Counter_Name'])] += 1
for k in sorted(t): print("%-14s %-26s %.4g" % (k[0], k[1], t[k] / n[k]))
for kk in ('k_narrow', 'k_pipe_solve', 'k_chain'):
    g = lambda c: t.get((kk, c), 0) / max(n.get((kk, c), 1), 1)
    if g('SQ_IFETCH'):
        print(kk, "avg ifetch latency %.0f cyc, vmem %.0f, smem %.0f, lds %.0f" % (g('SQ_IFETCH_LEVEL') / g('SQ_IFETCH'), g('SQ_INST_LEVEL_VMEM') / max(g('SQ_INSTS_VMEM'), 1), g('SQ_INST_LEVEL_SMEM') / max(g('SQ_INSTS_SMEM'), 1), g('SQ_INST_LEVEL_LDS') / max(g('SQ_INSTS_LDS'), 1)))
PY
done
