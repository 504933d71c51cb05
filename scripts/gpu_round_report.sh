#!/bin/bash
# Round report on the GPU box: tests, smoke, bench (faithful + throughput setting), rocprofv3 kernel stats and HBM PMC passes.
# Outputs land in gpurun_out/ (merged back by gpurun); summaries to keep are copied into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 900 python bench.py 2>&1 | tail -1 | tee $O/bench_default.json
timeout 400 python bench.py --fused --no-cpu-baseline --steps 100 2>&1 | tail -1 | tee $O/bench_fused.json
timeout 400 python bench.py --no-prefetch --no-cpu-baseline --steps 100 2>&1 | tail -1 | tee $O/bench_noprefetch.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --steps 40 --warmup 2 --no-cpu-baseline > $O/rocprof_stats.log 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_pmc; timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_pmc -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-prefetch > $O/rocprof_pmc_$c.log 2>&1
  python3 - <<PY
import csv, glob, collections, json
tot = collections.defaultdict(float); cnt = collections.defaultdict(set)
for f in glob.glob('/tmp/prof_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if k in ('k_order', 'k_pipe_begin', 'k_narrow', 'k_pipe_solve') and r['Counter_Name'] == '$c':
            tot[k] += float(r['Counter_Value']); cnt[k].add(r['Dispatch_Id'])
steps = max(1, len(cnt['k_order']))          # one k_order per control step (k_pipe_begin: one per env slice)
out = {k: {'dispatches': len(cnt[k]), 'KB_per_dispatch': tot[k] / max(1, len(cnt[k])), 'KB_per_step': tot[k] / steps} for k in tot}
out['steps'] = steps
print('$c', json.dumps(out))
json.dump(out, open('$O/pmc_$c.json', 'w'))
PY
done
head -12 $O/kernel_stats.csv
