timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -4
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --repeats 6 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default 20 steps:', round(d['value']), [round(v) for v in d['repeats']['values']], d['events'], d['config']['step_path'])"
timeout 300 python bench.py --steps 500 --warmup 10 --no-cpu-baseline --repeats 1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('500 steps:', round(d['value']), [round(v) for v in d['sustained']['per_100_steps_env_steps_per_s']], d['events'])"
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --repeats 1 --envs-per-gpu 131072 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('131072 envs:', round(d['value']), d['events'], d['config']['step_path'])"
