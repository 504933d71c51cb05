#!/bin/bash
# first contact of a chained-step build with the GPU: identity tests, then launch chains vs chained step.  Usage: gpu_chain_check.sh TAG
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "pipelined or chain" 2>&1 | tail -15 | tee $O/$1_identity.txt
for p in 1 2; do
  timeout 300 python bench.py --pipeline $p --steps 20 --warmup 5 --no-cpu-baseline --repeats 3 > $O/$1_bench_p$p.json 2> $O/$1_bench_p$p.err
  python3 - <<PY
import json
try:
    d = json.loads(open("$O/$1_bench_p$p.json").read().strip().splitlines()[-1])
    print("pipeline $p:", round(d["value"]), "env-steps/s", d["repeats"]["values"], d["events"], d["config"].get("step_path"))
except Exception as e:
    print("pipeline $p failed", e); print(open("$O/$1_bench_p$p.err").read()[-1500:])
PY
done
timeout 300 python bench.py --pipeline 2 --steps 500 --warmup 10 --no-cpu-baseline --repeats 1 > $O/$1_bench_p2_500.json 2> $O/$1_bench_p2_500.err
tail -c 600 $O/$1_bench_p2_500.json | head -c 600; echo
python3 -c "
import json; d=json.loads(open('$O/$1_bench_p2_500.json').read().strip().splitlines()[-1]); print('500 steps:', round(d['value']), d['sustained']['per_100_steps_env_steps_per_s'], d['events'])"
