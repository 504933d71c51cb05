"""BASELINE.json configs[0] on the GPU path: ONE env through task_suite.create_task_env (SingleEnvironment, the reference's numpy
observation dict), a 500-step random-action episode.  The only timing the reference has is the wall time of one env.step() of one env
(run_eval.py:103-124), so this reports exactly that: mean / p50 / p99 wall time per step() and per 500-step episode, for the launch
chains (pipeline 1, the batched default) and the fused single launch (pipeline 0), plus the bare device call (step_tensor, no numpy
conversion).  The oracle's single-thread figure for the same workload is bench.py's cpu_baseline (threads = 1)."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from so101_sim_amd import task_suite

out = {}
for name, pipeline in (("launch_chains", 1), ("fused", 0)):
    env = task_suite.create_task_env("SO100HandOverBanana", time_limit=10.0, random_state=0)
    env.sim.configure(pipeline=pipeline)
    spec = env.action_spec()
    rng = np.random.RandomState(1)
    t0 = time.perf_counter(); ts = env.reset(); t_reset = time.perf_counter() - t0
    for _ in range(20):                                    # warm-up (graph capture, lazy module loads)
        env.step(rng.uniform(spec.minimum, spec.maximum).astype(np.float32))
    ts = env.reset()
    per = []
    t_ep = time.perf_counter()
    n = 0
    while True:
        a = rng.uniform(spec.minimum, spec.maximum).astype(np.float32)
        t0 = time.perf_counter(); ts = env.step(a); per.append(time.perf_counter() - t0); n += 1
        if ts.last():
            break
    t_ep = time.perf_counter() - t_ep
    per = np.array(per) * 1e3
    # the bare device call: the action already a device tensor, one synchronize per step
    act = torch.zeros(1, 6, device=env.device)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200):
        env.step_tensor(act); torch.cuda.synchronize()
    bare = (time.perf_counter() - t0) / 200 * 1e3
    out[name] = dict(steps=n, episode_wall_s=t_ep, step_ms_mean=float(per.mean()), step_ms_p50=float(np.percentile(per, 50)),
                     step_ms_p99=float(np.percentile(per, 99)), env_steps_per_s=n / t_ep, reset_s=t_reset, bare_step_tensor_ms=bare,
                     info=env.sim.info())
    env.close()
print(json.dumps({"workload": "BASELINE.json configs[0]: SO100HandOverBanana, 1 env, SingleEnvironment.step(), 500-step random-action episode", **out}))
