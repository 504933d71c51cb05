#!/bin/bash
# chained step: throughput + where the persistent wavefronts' time goes.  Usage: gpu_chain_stats.sh TAG [bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R; T=$1; shift
timeout 300 python bench.py --pipeline 2 --steps 20 --warmup 5 --no-cpu-baseline --repeats 3 "$@" > $O/${T}_bench.json 2> $O/${T}_bench.err || tail -20 $O/${T}_bench.err
python3 - <<PY
import json
d = json.loads(open("$O/${T}_bench.json").read().strip().splitlines()[-1])
cs = d["config"]["chain_stats"]; n = d["config"]["envs_per_gpu"]
print("value", round(d["value"]), [round(v) for v in d["repeats"]["values"]], "ms/step", round(d["ms_per_step"], 3))
steps = d["steps"] * d["repeats"]["n"]
life = cs["t_life"]
print("per wave-second shares: pop %.3f idle %.3f narrow %.3f solve %.3f (life %.2f wave-s over %d waves)" % (cs["t_pop"]/life, cs["t_idle"]/life, cs["t_narrow"]/life, cs["t_solve"]/life, life, cs["waves"]))
print("items: narrow %d (%.1f us each) solve %d (%.1f us each) idle rounds %d; pop %.2f us per item" % (cs["n_narrow"], 1e6*cs["t_narrow"]/max(cs["n_narrow"],1), cs["n_solve"], 1e6*cs["t_solve"]/max(cs["n_solve"],1), cs["n_idle"], 1e6*cs["t_pop"]/max(cs["n_narrow"]+cs["n_solve"],1)))
nn, ns = max(cs["n_narrow"],1), max(cs["n_solve"],1)
print("narrow us: load %.1f pairs %.1f finish %.1f | solve us: load+gather %.1f compute %.1f store+broadphase %.1f publish %.1f" % (
  1e6*cs["t_narrow_load"]/nn, 1e6*cs["t_narrow_pairs"]/nn, 1e6*cs["t_narrow_finish"]/nn,
  1e6*cs["t_solve_load_gather"]/ns, 1e6*cs["t_solve_compute"]/ns, 1e6*cs["t_solve_store_broad"]/ns, 1e6*cs["t_solve_publish"]/ns))
print(d["events"])
PY
