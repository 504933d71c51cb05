"""Throughput of the ALOHA hand-over env on the general-tree engine (run through gpurun): env-steps/s of so101_tree_step with
uniform random joint targets around the home pose, at a few batch sizes.  The env's default step path (batches: the launch chain of so101_tree.hpp).
    python scripts/gpu_aloha_bench.py [banana|pen] [n_envs ...] > gpurun_out/r03_aloha_bench.json"""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from so101_sim_amd import task_suite          # noqa: E402
from so101_sim_amd.model import scenes        # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "banana"
out = []
for n in ([int(x) for x in sys.argv[2:]] or (256, 1024, 4096)):
    env = task_suite.create_task_env("HandOverBanana" if name == "banana" else "HandOverPen", time_limit=10.0, random_state=0, n_envs=n)
    t0 = time.time(); env.reset(); torch.cuda.synchronize(); t_reset = time.time() - t0
    home = torch.tensor(np.concatenate([scenes.ALOHA_HOME_CTRL] * 2), dtype=torch.float32, device=env.device)
    g = torch.Generator(device=env.device); g.manual_seed(1)
    spec = env.action_spec()
    lo, hi = torch.tensor(spec.minimum, device=env.device), torch.tensor(spec.maximum, device=env.device)
    steps = 20
    acts = [torch.clamp(home + 0.5 * (torch.rand(n, 14, generator=g, device=env.device) - 0.5), lo, hi) for _ in range(steps + 3)]
    for k in range(3):
        env.step_tensor(acts[k])
    torch.cuda.synchronize(); t0 = time.time()
    for k in range(steps):
        env.step_tensor(acts[3 + k])
    torch.cuda.synchronize(); dt = time.time() - t0
    d = env.diagnostics().cpu().numpy()
    out.append({"workload": f"HandOver{name.capitalize()} (ALOHA, nq 30 / nv 28 / nu 14), {n} envs, random joint targets around the home pose",
                "env_steps_per_s": n * steps / dt, "ms_per_step": 1e3 * dt / steps, "reset_s": t_reset,
                "mean_contacts": float(d[:, 0].mean()), "mean_rows": float(d[:, 1].mean()), "mean_newton_iterations": float(d[:, 2].mean()),
                "flagged_envs": int((d[:, 4] != 0).sum())})
    env.close()
    print(json.dumps(out[-1]), flush=True)
