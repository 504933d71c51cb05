import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so101_sim_amd import task_suite, pregrasp
from so101_sim_amd.model import scenes
from oracle.oracle import Oracle
from tests.simharness import ArraySim
from tests import parity_cases as pc
raw32, _ = scenes.load_blob("banana", "f32"); raw64, _ = scenes.load_blob("banana", "f64")
cwd = os.getcwd(); os.chdir("/tmp")
n = 64
env = task_suite.create_task_env("SO100HandOverBanana", time_limit=10.0, random_state=0, n_envs=n)
PQ, PV, PC = (t.cpu().numpy().astype(np.float64) for t in pregrasp.build_pickplace_pool(env, pool_size=n, seed=3))
env.close(); os.chdir(cwd)
np.savez("gpurun_out/pool_debug2.npz", PQ=PQ, PV=PV, PC=PC)
half = n // 2
idx = list(range(half, half + 8))
sim = ArraySim(raw32, len(idx), backend="gpu", last_step=500)
sim.set_state(PQ[:, idx], PV[:, idx], PC[:, idx], np.zeros((18, len(idx))))
dbg = sim.debug_forward()
sim.begin_episode()
act = PC[:, idx].T.astype(np.float32).copy(); act[:, 5] -= 0.3
for sub in range(1, 11):
    pass
obs, rew, disc, st = sim.step(act)
q1, v1, _ = sim.get_state()
for j, k in enumerate(idx):
    o = Oracle(raw64); o.env_config(seed=0, env_id=j, last_step=500)
    o.set_state(PQ[:, k], PV[:, k], np.zeros(18)); o.forward()
    problems, _, _ = pc._compare_contact_lists(dbg[j]["contacts"], o.contacts())
    a = o.qacc()[0]
    print("entry", k, "forward: ncon", dbg[j]["ncon"], len(o.contacts()), "problems", problems[:2], "qacc err %.2e" % (np.abs(dbg[j]["qacc"] - a).max() / np.abs(a).max()))
    o.env_begin(); o.env_step(act[j].astype(np.float64)); qo, vo, _ = o.get_state()
    dq = np.abs(q1[:, j] - qo); dv = np.abs(v1[:, j] - vo)
    print("    step: dq max %.2e at %d, dv max %.2e at %d" % (dq.max(), dq.argmax(), dv.max(), dv.argmax()), "bowl q", np.round(PQ[13:20, k], 4))
