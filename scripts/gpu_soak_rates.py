"""Failure rates over long random-action rollouts (run through gpurun): physics errors, overflows and unsettled resets per env-step for the
SO100 hand-over with the default (EPA) narrowphase and the MPR option, and physics errors of the ALOHA hand-over on the general-tree engine.
    python scripts/gpu_soak_rates.py > gpurun_out/r03_soak_rates.json"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.chdir("/tmp")                          # (no calibration file: offsets off, as in bench.py)
from so101_sim_amd import task_suite      # noqa: E402

out = []
N, STEPS = 4096, 3000
for narrow in ("epa", "mpr"):
    env = task_suite.create_task_env("SO100HandOverBanana", time_limit=10.0, random_state=0, n_envs=N, narrowphase=narrow)
    spec = env.action_spec()
    lo, hi = torch.tensor(spec.minimum, device=env.device), torch.tensor(spec.maximum, device=env.device)
    g = torch.Generator(device=env.device); g.manual_seed(1)
    env.reset(); env.events(clear=True)
    t0 = time.time()
    for k in range(STEPS):
        env.step_tensor(lo + (hi - lo) * torch.rand(N, 6, generator=g, device=env.device))
    torch.cuda.synchronize(); dt = time.time() - t0
    ev = env.events()
    out.append({"workload": f"SO100HandOverBanana, {N} envs x {STEPS} steps (six 500-step episodes), uniform random actions, narrowphase {narrow}",
                "env_steps_per_s": N * STEPS / dt, "events_per_env_step": {k: v / (N * STEPS) for k, v in ev.items()}, "events": ev})
    env.close()
    print(json.dumps(out[-1]), flush=True)
N, STEPS = 2048, 1000
env = task_suite.create_task_env("HandOverBanana", time_limit=10.0, random_state=0, n_envs=N)
spec = env.action_spec()
lo, hi = torch.tensor(spec.minimum, device=env.device), torch.tensor(spec.maximum, device=env.device)
g = torch.Generator(device=env.device); g.manual_seed(1)
env.reset()
errors = early = flagged = 0
t0 = time.time()
for k in range(STEPS):
    obs, r, d, st = env.step_tensor(lo + (hi - lo) * torch.rand(N, 14, generator=g, device=env.device))
    last = st == 2
    errors += int((last & (d == 0) & (r == 0)).sum())
    fl = env.diagnostics()[:, 4]
    flagged += int(((fl & 7) != 0).sum())
torch.cuda.synchronize(); dt = time.time() - t0
out.append({"workload": f"HandOverBanana (ALOHA, general-tree engine), {N} envs x {STEPS} steps (two 500-step episodes), uniform random actions over the action spec",
            "physics_errors_per_env_step": errors / (N * STEPS), "overflow_flags_per_env_step": flagged / (N * STEPS),
            "note": "rates only: this loop reads the diagnostics back every step and resets inside the step calls, so its wall time is not a throughput"})
print(json.dumps(out[-1]), flush=True)
