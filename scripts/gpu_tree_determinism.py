import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from tests.test_tree_parity import TreeArraySim, _blobs, scenes
raw32 = _blobs("banana")[1]
import os
NSUB=int(os.environ.get('NSUB','10'))
def run(pipeline, n=64, steps=int(os.environ.get('STEPS','4'))):
    sim = TreeArraySim(raw32, n, backend="gpu")
    sim.enable_env(seed=5, last_step=1000, settle_max_substeps=300, pipeline=pipeline, n_substeps=NSUB, reward_mode=0)
    rng = np.random.RandomState(2); tr = []
    for k in range(steps):
        a = np.tile(np.concatenate([scenes.ALOHA_HOME_CTRL] * 2), (n, 1)) + 0.2 * rng.normal(size=(n, 14))
        obs, r, d, st = sim.step(a)
        tr.append((obs.copy(), [x.copy() for x in sim.get_state()], sim.get_diag().copy()))
    return tr
A = run(0); C = run(1)
def cmp(X, Y, name):
    for k, (x, y) in enumerate(zip(X, Y)):
        so = not np.array_equal(x[0], y[0]); ss = [not np.array_equal(a, b) for a, b in zip(x[1], y[1])]; sd = not np.array_equal(x[2], y[2])
        print(name, "step", k, "obs differ", so, "state differ", ss, "diag differ", sd)
        if sd:
            bad = np.where((x[2] != y[2]).any(axis=1))[0]
            print("   envs", bad[:8], "diag", x[2][bad[0]], y[2][bad[0]])
        if any(ss):
            for a, b in zip(x[1], y[1]):
                if not np.array_equal(a, b):
                    bad = np.where((a != b).reshape(a.shape[0], -1).any(axis=1))[0]; print("   state envs", bad[:8], "max abs diff", np.abs(a - b).max())
cmp(A, C, "single vs chain")
