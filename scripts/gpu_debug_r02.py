import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from so101_sim_amd.model import scenes
from oracle.oracle import Oracle
from tests.simharness import ArraySim
raw32, _ = scenes.load_blob("banana", "f32"); raw64, _ = scenes.load_blob("banana", "f64")
n, seed, settle, last_step, iterations = 4, 11, 200, 7, 50
sim = ArraySim(raw32, n, backend="gpu", seed=seed, settle_max_substeps=settle, last_step=last_step, solver_iterations=iterations, env_id_base=100)
sim.reset()
q0, v0, _ = sim.get_state()
oracles = []
for e in range(n):
    o = Oracle(raw64); o.set_solver(iterations, -1.0)
    o.env_config(seed=seed, env_id=100 + e, last_step=last_step, settle_max_substeps=settle)
    o.env_reset(); oracles.append(o)
    qo, vo, _ = o.get_state()
    print("reset env", e, "dq", np.abs(q0[:, e] - qo).max(), "dv", np.abs(v0[:, e] - vo).max())
rng = np.random.RandomState(seed)
for t in range(1, 8):
    act = rng.uniform(-0.4, 0.4, size=(n, 6)).astype(np.float32)
    obs, rew, disc, st = sim.step(act)
    q1, v1, _ = sim.get_state()
    for e, o in enumerate(oracles):
        oo, orew, odisc, ost = o.env_step(act[e].astype(np.float64))
        qo, vo, _ = o.get_state()
        arm = [(c["geom1"], c["geom2"], round(c["dist"], 6)) for c in o.contacts() if 1 <= c["geom1"] <= 18 or 1 <= c["geom2"] <= 18]
        print(" t", t, "env", e, "dq arm %.2e" % np.abs(q1[:6, e] - qo[:6]).max(), "dq props %.2e" % np.abs(q1[6:, e] - qo[6:]).max(), "diag", sim.get_diag()[e][:5], "arm contacts", arm[:6])
