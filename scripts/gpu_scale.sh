#!/bin/bash
# Scaling table for whoever has an 8-GPU MI355X node (the builder's boxes have one GPU: NO scaling curve has been measured, profiles/README.md).
#   scripts/gpu_scale.sh [envs-per-gpu=32768] [steps=30] [list of N="1 2 4 8"]   ->  gpurun_out/scale_<envs>.json
# Every N is one `bench.py --gpus N` run (it starts its own N ranks, one per GPU, RCCL process group, weak scaling: envs-per-gpu fixed,
# env ids sharded contiguously, all-gather of returns for logging only).  The output holds each run's full JSON line - value, per-rank
# ms_per_step and the process-group evidence (`dist`) - plus value(N) / (N x value(1)).
ENVS=${1:-32768}; STEPS=${2:-30}; NS=${3:-"1 2 4 8"}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=${HSA_ENABLE_IPC_MODE_LEGACY:-0}
have=$(python3 -c "import torch; print(torch.cuda.device_count())")
lines=()
for n in $NS; do
  if [ "$n" -gt "$have" ]; then echo "skipping N=$n: $have GPU(s) visible" >&2; continue; fi
  f=/tmp/scale_$n.json
  timeout 1200 python3 bench.py --gpus $n --envs-per-gpu $ENVS --steps $STEPS --warmup 5 > $f 2>/tmp/scale_$n.err || { echo "N=$n failed: $(tail -3 /tmp/scale_$n.err)" >&2; continue; }
  lines+=("$f")
done
python3 - "$ENVS" "${lines[@]}" > $O/scale_$ENVS.json <<'PY'
import json, sys
envs, files = int(sys.argv[1]), sys.argv[2:]
runs = [json.loads(open(f).read().strip().splitlines()[-1]) for f in files]
base = next((r["value"] for r in runs if r["n_gpus"] == 1), None)
print(json.dumps({"envs_per_gpu": envs, "scaling": "weak",
                  "table": [{"n_gpus": r["n_gpus"], "value": r["value"], "ms_per_step": r["ms_per_step"],
                             "ratio_to_n_times_single": (r["value"] / (r["n_gpus"] * base)) if base else None,
                             "dist": r.get("dist")} for r in runs],
                  "runs": runs}, indent=1))
PY
python3 -c "
import json; d = json.load(open('$O/scale_$ENVS.json'))
for t in d['table']: print(t['n_gpus'], 'GPU(s):', round(t['value']), 'env-steps/s', 'ratio', t['ratio_to_n_times_single'])
"
