"""Which envs make k_pipe_solve long: solver time of the last substep vs Newton iterations and contact count."""
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
d = s.get_diag().astype(np.int64)
ncon, nefc, it, ts = d[:, 0], d[:, 1], d[:, 2], d[:, 6] * 1e-2
print("iterations histogram:", np.bincount(np.minimum(it, 20)))
print("solver us by iterations:", {int(k): round(float(ts[it == k].mean()), 1) for k in np.unique(it)})
k = np.argsort(ts)[-20:][::-1]
print("top 20: us", np.round(ts[k]), "iters", it[k], "ncon", ncon[k], "nefc", nefc[k], "flags", d[k, 4])
for lo_, hi_ in ((0, 4), (4, 8), (8, 12), (12, 16), (16, 24), (24, 33)):
    m = (ncon >= lo_) & (ncon < hi_)
    if m.any():
        print("ncon [%2d,%2d): %4d envs, iters %.2f, solver %.1f us, us/iter %.1f" % (lo_, hi_, m.sum(), it[m].mean(), ts[m].mean(), ts[m].sum() / max(1, it[m].sum())))
