#!/bin/bash
# Instruction-cache / issue counters of the general-tree engine's control step - the launch chain k_tree_pipe_begin, k_tree_narrow,
# k_tree_pipe_solve at 4096 ALOHA envs, summed per control step (separate --pmc passes, kernel-trace only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
HASH=$(cd $R && python3 -c "from so101_sim_amd import build; print(build.source_hash())")
for c in FETCH_SIZE WRITE_SIZE "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU" "SQ_IFETCH SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS" "SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"; do
  n=$(echo $c | cut -d' ' -f1)
  rm -rf /tmp/pt_$n; timeout 600 rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pt_$n -- python3 $R/scripts/gpu_aloha_bench.py banana 4096 > $O/tree_pmc_$n.log 2>&1
done
python3 - $HASH $O <<'PY'
import csv, glob, collections, json, sys
KERN = ("k_tree_pipe_begin", "k_tree_narrow", "k_tree_pipe_solve")
t = collections.defaultdict(float); per = collections.defaultdict(lambda: collections.defaultdict(float)); begins = collections.defaultdict(int)
for f in glob.glob('/tmp/pt_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        for kk in KERN:
            if k.endswith(kk):
                t[r['Counter_Name']] += float(r['Counter_Value']); per[kk][r['Counter_Name']] += float(r['Counter_Value'])
                if kk == "k_tree_pipe_begin": begins[r['Counter_Name']] += 1
SLICES = 4                                            # env slices of a 4096-env step (tu_tree.hip): one k_tree_pipe_begin each
steps = {c: max(1, begins[c] // SLICES) for c in t}
m = {c: t[c] / steps[c] for c in t}
for kk in KERN:
    for c in sorted(per[kk]): print("%-18s %-30s %.4g per control step" % (kk, c, per[kk][c] / steps[c]))
for c in sorted(m): print("control step (4096 envs)  %-30s %.4g (%d steps)" % (c, m[c], steps[c]))
# what bench.py --workload aloha reports as roofline.traffic / roofline.compute (FETCH_SIZE doubled: gfx950 tallies 128-B requests at 64 B; KB)
out = {"build": sys.argv[1], "workload": "scripts/gpu_aloha_bench.py banana 4096: k_tree_pipe_begin + k_tree_narrow + k_tree_pipe_solve dispatches of a control step of 4096 envs, mean over the steps",
       "steps": max(steps.values()) if steps else 0, "per_step": m, "per_kernel": {kk: {c: per[kk][c] / steps[c] for c in per[kk]} for kk in KERN},
       "hbm_bytes_per_step": (2 * m.get("FETCH_SIZE", 0.0) + m.get("WRITE_SIZE", 0.0)) * 1024.0,
       "valu_insts_per_step": m.get("SQ_INSTS_VALU"), "salu_insts_per_step": m.get("SQ_INSTS_SALU"),
       "wait_fraction": m["SQ_WAIT_ANY"] / m["SQ_WAVE_CYCLES"] if m.get("SQ_WAVE_CYCLES") else None,
       "active_lane_fraction": m["SQ_THREAD_CYCLES_VALU"] / (64.0 * m["SQ_ACTIVE_INST_VALU"]) if m.get("SQ_ACTIVE_INST_VALU") else None}
json.dump(out, open(sys.argv[2] + "/pmc_tree_" + sys.argv[1] + ".json", "w"), indent=1)
PY
