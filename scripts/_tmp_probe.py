import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so101_sim_amd import pregrasp, task_suite
from so101_sim_amd.model import scenes
from tests.simharness import ArraySim
from tests import parity_cases as pc
from oracle.oracle import Oracle
os.chdir("/tmp")
raw32, _ = scenes.load_blob("banana", "f32"); raw64, _ = scenes.load_blob("banana", "f64")
for seed in (3, 4, 5):
    n = 64
    env = task_suite.create_task_env("SO100HandOverBanana", time_limit=10.0, random_state=0, n_envs=n)
    PQ, PV, PC = (t.cpu().numpy().astype(np.float64) for t in pregrasp.build_pickplace_pool(env, pool_size=n, seed=seed))
    env.close()
    idx = list(range(0, 32))
    sim = ArraySim(raw32, len(idx), backend="gpu", last_step=500)
    sim.set_state(PQ[:, idx], PV[:, idx], PC[:, idx], np.zeros((18, len(idx))))
    dbg = sim.debug_forward()
    diff = []
    for j, k in enumerate(idx):
        o = Oracle(raw64); o.set_state(PQ[:, k], PV[:, k], np.zeros(18)); o.set_ctrl(PC[:, k]); o.forward()
        problems, _, _ = pc._compare_contact_lists(dbg[j]["contacts"], o.contacts())
        diff.append(int(bool(problems)))
    print("seed", seed, "differs first 8:", sum(diff[:8]), "first 16:", sum(diff[:16]), "all 32:", sum(diff), diff)
