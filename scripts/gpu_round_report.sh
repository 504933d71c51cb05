#!/bin/bash
# Round report on the GPU box: tests, smoke, bench (faithful + throughput setting), rocprofv3 kernel stats and HBM PMC passes.
# Outputs land in gpurun_out/ (merged back by gpurun); summaries to keep are copied into profiles/ by hand.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 600 python -m pytest tests -m gpu -q 2>&1 | tail -5 | tee $O/pytest_gpu.txt
timeout 200 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 900 python bench.py 2>&1 | tail -1 | tee $O/bench_default.json
timeout 400 python bench.py --solver-iterations 10 --no-cpu-baseline 2>&1 | tail -1 | tee $O/bench_10it.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_stats; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --steps 40 --warmup 2 --no-cpu-baseline > $O/rocprof_stats.log 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/prof_pmc; timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_pmc -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline > $O/rocprof_pmc_$c.log 2>&1
  python3 - <<PY
import csv, glob
vals=[]
for f in glob.glob('/tmp/prof_pmc/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_step' in r['Kernel_Name'] and r['Counter_Name']=='$c':
            vals.append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
import collections
d=collections.defaultdict(float)
for k,v in vals: d[k]+=v
xs=sorted(d.items())
print('$c per k_step dispatch (KB as reported):', [round(v,1) for k,v in xs][:14])
open('$O/pmc_$c.txt','w').write(repr(xs))
PY
done
head -12 $O/kernel_stats.csv
