"""Stage clocks of k_pipe_solve (so101_debug_stages) on the bench workload."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so101_sim_amd.model import scenes
from tests.simharness import ArraySim
raw32, meta = scenes.load_blob("banana", "f32")
N = 4096
s = ArraySim(raw32, N, backend="gpu", seed=0, settle_max_substeps=300, last_step=100000, prefetch_resets=0)
s.reset()
lo = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0], dtype=np.float32)
hi = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08], dtype=np.float32)
rng = np.random.RandomState(2)
for t in range(30):
    s.step(rng.uniform(lo, hi, size=(N, 6)).astype(np.float32))
st = torch.zeros(N, 8, dtype=torch.int32, device=s.dev)
s.sim.debug_stages(st.data_ptr()); torch.cuda.synchronize()
st = st.cpu().numpy().astype(np.int64)
names = ["smooth", "gather", "constraints", "solver", "integrate", "next broadphase"]
us = st[:, :6] * 1e-2
print("per-env stage time of one substep (us): mean / p50 / p99 / max")
for i, n in enumerate(names):
    print("  %-16s %7.1f %7.1f %7.1f %7.1f" % (n, us[:, i].mean(), np.median(us[:, i]), np.percentile(us[:, i], 99), us[:, i].max()))
tot = us.sum(1)
print("  %-16s %7.1f %7.1f %7.1f %7.1f" % ("total", tot.mean(), np.median(tot), np.percentile(tot, 99), tot.max()))
print("sum over envs / 1792 resident waves = %.3f ms ; max env %.3f ms" % (tot.sum() * 1e-3 / 1792, tot.max() * 1e-3))
ncon, it = st[:, 6], st[:, 7]
for lo_, hi_ in ((0, 1), (1, 5), (5, 9), (9, 13), (13, 20), (20, 33)):
    mk = (ncon >= lo_) & (ncon < hi_)
    if mk.any():
        print("  ncon [%2d,%2d): %4d envs  constraints %.1f  solver %.1f us  iters %.1f" % (lo_, hi_, mk.sum(), us[mk, 2].mean(), us[mk, 3].mean(), it[mk].mean()))
