"""Per-env stage clocks (diag words 5-7) of k_step on the bench workload: is the launch bound by throughput
(sum of env times / resident waves) or by its slowest env (tail)?"""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so101_sim_amd.model import scenes
from tests.simharness import ArraySim

raw32, _ = scenes.load_blob("banana", "f32")
N = int(os.environ.get("N", "4096"))
s = ArraySim(raw32, N, backend="gpu", seed=0, settle_max_substeps=300, last_step=100000)
s.reset()
lo = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0], dtype=np.float32)
hi = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08], dtype=np.float32)
rng = np.random.RandomState(2)
for t in range(int(os.environ.get("ROLL", "40"))):
    act = rng.uniform(lo, hi, size=(N, 6)).astype(np.float32)
    s.action.copy_(torch.from_numpy(act).to(s.dev)) if False else None
    torch.cuda.synchronize(); t0 = time.time()
    s.step(act)
    torch.cuda.synchronize(); wall = (time.time() - t0) * 1e3
    if t % 5 == 4:
        d = s.get_diag().astype(np.int64)
        coll, solve, tot = d[:, 5] * 1e-5, d[:, 6] * 1e-5, d[:, 7] * 1e-5       # ms
        k = np.argsort(tot)[-3:]
        print("t=%d wall %.1f ms | per-env total: mean %.2f p50 %.2f p99 %.2f max %.2f | collision mean %.2f max %.2f | solve mean %.2f max %.2f | sum/1536 %.1f ms" % (
            t, wall, tot.mean(), np.median(tot), np.percentile(tot, 99), tot.max(), coll.mean(), coll.max(), solve.mean(), solve.max(), tot.sum() / 1536))
        print("    slowest envs: ncand", d[k, 3], "ncon", d[k, 0], "tot", np.round(tot[k], 2), "coll", np.round(coll[k], 2), "solve", np.round(solve[k], 2), "iters", d[k, 2])
        # cost model: collision time vs candidates
        nc = d[:, 3]
        for lo_, hi_ in ((0, 4), (4, 8), (8, 16), (16, 32), (32, 64), (64, 999)):
            mk = (nc >= lo_) & (nc < hi_)
            if mk.any():
                print("    ncand [%d,%d): %d envs, collision %.3f ms/env, %.1f us/candidate" % (lo_, hi_, mk.sum(), coll[mk].mean(), 1e3 * coll[mk].sum() / max(1, 10 * nc[mk].sum())))
