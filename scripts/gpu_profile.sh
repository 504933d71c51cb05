#!/bin/bash
# Round profile on the GPU box (run through gpurun): rocprofv3 kernel stats of the default bench command, then separate
# --pmc passes (kernel-trace only) for HBM traffic and SQ counters, summarised into profiles-ready files under gpurun_out/.
#   gpurun_out/<tag>_kernel_stats.csv      rocprofv3 --kernel-trace --stats
#   gpurun_out/pmc_<build hash>.json       what bench.py reads as roofline.traffic / roofline.compute
# FETCH_SIZE is doubled per the gfx950 note of MI355X_MICROARCH.md (128-B requests tallied at 64 B); FETCH and WRITE need
# separate passes (TCC slots); SQ counters a third.  Units: FETCH_SIZE / WRITE_SIZE in KB.
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
HASH=$(cd $R && python3 -c "from so101_sim_amd import build; print(build.source_hash())")
cd /tmp && export TMPDIR=/tmp
STEPS=30
rm -rf /tmp/prof_stats; timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_stats -- python3 $R/bench.py --steps $STEPS --warmup 3 --no-cpu-baseline > $O/${TAG}_rocprof_stats.log 2>&1
find /tmp/prof_stats -name "*kernel_stats.csv" -exec cp {} $O/${TAG}_kernel_stats.csv \;
for c in FETCH_SIZE WRITE_SIZE "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/prof_pmc_$n; timeout 900 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/prof_pmc_$n -- python3 $R/bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-prefetch > $O/${TAG}_rocprof_pmc_$n.log 2>&1
done
python3 - <<PY
import csv, glob, collections, json
kern = ('k_order', 'k_pipe_begin', 'k_narrow', 'k_pipe_solve')
tot = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for f in glob.glob('/tmp/prof_pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        for kk in kern:
            if k.startswith(kk) or (' ' + kk) in k:
                tot[kk][r['Counter_Name']] += float(r['Counter_Value']); disp[(kk, r['Counter_Name'])].add(r['Dispatch_Id'])
steps = max(1, len(disp[('k_order', 'FETCH_SIZE')]))          # one k_order per control step
per_kernel = {}
for kk in kern:
    per_kernel[kk] = {c: v / max(1, len(disp[(kk, c)])) for c, v in tot[kk].items()}
    per_kernel[kk]['dispatches_per_step'] = len(disp[(kk, 'FETCH_SIZE')]) / steps
def per_step(c):
    return sum(tot[kk][c] for kk in kern) / max(1, len(disp[('k_order', c)]))
fetch_kb, write_kb = per_step('FETCH_SIZE'), per_step('WRITE_SIZE')
valu, salu = per_step('SQ_INSTS_VALU'), per_step('SQ_INSTS_SALU')
wc, wa = per_step('SQ_WAVE_CYCLES'), per_step('SQ_WAIT_ANY')
thr, act = per_step('SQ_THREAD_CYCLES_VALU'), per_step('SQ_ACTIVE_INST_VALU')
out = {'build': '$HASH', 'workload': 'bench.py default (SO100HandOverBanana, 4096 envs), --no-prefetch for the PMC passes, 12 timed steps',
       'steps_sampled': steps,
       'hbm_bytes_per_step': (2 * fetch_kb + write_kb) * 1024, 'fetch_kb_per_step_raw': fetch_kb, 'write_kb_per_step': write_kb,
       'valu_insts_per_step': valu, 'salu_insts_per_step': salu,
       'wait_fraction': wa / wc if wc else None,
       'active_lane_fraction': (thr / (act * 64)) if act else None,
       'per_kernel_per_dispatch': per_kernel,
       'note': 'FETCH_SIZE doubled (gfx950: 128-B requests tallied at 64 B); SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles; '
               'active_lane_fraction = SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU)'}
json.dump(out, open('$O/pmc_$HASH.json', 'w'), indent=1)
print(json.dumps({k: out[k] for k in ('build', 'hbm_bytes_per_step', 'valu_insts_per_step', 'wait_fraction', 'active_lane_fraction')}))
for kk in kern:
    print(kk, {c: round(v) for c, v in per_kernel[kk].items()})
PY
head -12 $O/${TAG}_kernel_stats.csv
