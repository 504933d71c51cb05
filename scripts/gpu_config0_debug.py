"""configs[0] single env, 500 random steps (the test's loop): the worst one-step difference against the oracle among the steps without arm
contact, then that step again substep by substep (kernel state -> oracle state each substep) with the contact lists of both."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from so101_sim_amd import task_suite
from so101_sim_amd.model import scenes
from oracle.oracle import Oracle
from tests import parity_cases as pc
from tests.simharness import ArraySim
raw32, meta = scenes.load_blob("banana", "f32"); raw64, _ = scenes.load_blob("banana", "f64")
gn = meta["geom_names"]
os.chdir(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
env = task_suite.create_task_env("SO100HandOverBanana", time_limit=10.0, random_state=0)
spec = env.action_spec(); ts = env.reset()
rng = np.random.RandomState(0)
o1 = Oracle(raw64); rows = []
for t in range(1, 501):
    a = rng.uniform(spec.minimum, spec.maximum).astype(np.float32)
    before = [x[:, 0].cpu().numpy().astype(np.float64) for x in (env.qpos, env.qvel, env.warm)]
    ts = env.step(a)
    if ts.last(): break
    o1.set_state(*before); o1.set_ctrl(a.astype(np.float64)); o1.substeps(10)
    q1, v1, _ = o1.get_state()
    arm = any(pc._arm_geom(c["geom1"]) or pc._arm_geom(c["geom2"]) for c in o1.contacts())
    ps = ts.observation["physics_state"]
    rows.append((t, np.abs(ps[:20] - q1).max(), np.abs(ps[20:] - v1).max(), arm, before, a))
free = [r for r in rows if not r[3]]
free.sort(key=lambda r: -r[1])
print("free steps", len(free), "worst five:", [(r[0], float(r[1]), float(r[2])) for r in free[:5]])
t, dq, dv, _, before, a = free[0]
sim = ArraySim(raw32, 1, backend="gpu", seed=0, last_step=100000, prefetch_resets=0)
o = Oracle(raw64)
q, v, w = before
for k in range(10):
    sim.set_state(q[:, None], v[:, None], a.astype(np.float64)[:, None], w[:, None])
    d = sim.debug_forward()[0]
    o.set_state(q, v, w); o.set_ctrl(a.astype(np.float64)); o.forward()
    ref = o.contacts()
    problems, tot, loose, wit = pc._compare_contact_lists(d["contacts"], ref)
    sim.physics(1)
    qg, vg, wg = [x[:, 0] for x in sim.get_state()]
    o.substeps(1)
    qo, vo, wo = o.get_state()
    i = int(np.argmax(np.abs(qg - qo)))
    print("substep", k, "contacts", len(ref), "problems", [(gn[int(p.split(',')[0][1:])], gn[int(p.split(',')[1].split(')')[0])], p.split(':')[1]) for p in problems], "loose", loose, "witness", wit,
          "max dq %.2e at %d  max dv %.2e" % (np.abs(qg - qo).max(), i, np.abs(vg - vo).max()))
    q, v, w = qg.astype(np.float64), vg.astype(np.float64), wg.astype(np.float64)
