"""Attributes k_physics time to stages on the real workload: 4096 envs are rolled out with random actions, then the
SAME state is stepped 10 substeps with stages masked off (SO101_DEBUG_PHASES: bit0 collision, bit1 constraint rows +
warm start, bit2 PGS iterations)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so101_sim_amd.model import scenes
from tests.simharness import ArraySim

raw32, _ = scenes.load_blob("banana", "f32")
N = int(os.environ.get("N", "4096"))
iters = int(os.environ.get("ITERS", "100"))
s = ArraySim(raw32, N, backend="gpu", seed=0, solver_iterations=iters, settle_max_substeps=300, last_step=100000)
s.reset()
lo = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0], dtype=np.float32)
hi = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08], dtype=np.float32)
rng = np.random.RandomState(2)
for t in range(int(os.environ.get("ROLL", "30"))):
    s.step(rng.uniform(lo, hi, size=(N, 6)).astype(np.float32))
q0, v0, w0 = s.get_state(); c0 = s._get(s.ctrl)
d = s.get_diag()
print("state: ncon mean %.1f max %d, nefc mean %.1f, iters mean %.1f, ncand mean %.1f" % (d[:, 0].mean(), d[:, 0].max(), d[:, 1].mean(), d[:, 2].mean(), d[:, 3].mean()))
arm = [(d[:, 0] > k).mean() for k in (8, 12, 16, 24)]
print("fraction of envs with ncon > 8/12/16/24:", arm)
for ph in (0, 1, 3, 7):
    os.environ["SO101_DEBUG_PHASES"] = str(ph)
    ts = []
    for rep in range(3):
        s.set_state(q0, v0, c0, w0); torch.cuda.synchronize()
        t = time.time(); s.physics(10); torch.cuda.synchronize(); ts.append(time.time() - t)
    print(json.dumps(dict(phases=ph, iters=iters, ms_per_control_step=min(ts) * 1e3)))
