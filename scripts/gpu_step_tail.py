"""Per-step wall time of k_step against the per-env diagnostics (tail analysis)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so101_sim_amd.model import scenes
from tests.simharness import ArraySim
raw32, _ = scenes.load_blob("banana", "f32")
N = 4096
s = ArraySim(raw32, N, backend="gpu", seed=0, settle_max_substeps=300, last_step=100000, solver_iterations=int(os.environ.get("ITERS", "0")))
s.reset()
lo = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0], dtype=np.float32)
hi = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08], dtype=np.float32)
rng = np.random.RandomState(2)
for t in range(int(os.environ.get("STEPS", "80"))):
    a = rng.uniform(lo, hi, size=(N, 6)).astype(np.float32)
    s._put(s.action, a); torch.cuda.synchronize()
    t0 = time.time()
    p = s.ptr
    s.sim.step(p(s.action), p(s.obs), p(s.reward_), p(s.discount), p(s.step_type), s.stream()); torch.cuda.synchronize()
    dt = time.time() - t0
    d = s.get_diag()
    if t % 4 == 0 or dt > 0.06:
        it = d[:, 2]
        print("t=%3d %.1f ms | iters mean %.1f p99 %d max %d (>20: %d) | ncon mean %.1f max %d | ncand max %d | flags div %d conovf %d" % (
            t, dt * 1e3, it.mean(), np.percentile(it, 99), it.max(), (it > 20).sum(), d[:, 0].mean(), d[:, 0].max(), d[:, 3].max(), ((d[:, 4] & 8) != 0).sum(), ((d[:, 4] & 2) != 0).sum()))
