"""How often is a free body's pose BIT-IDENTICAL from one control step (one substep) to the next?  (Round 6: a narrowphase record of a
pair whose two bodies did not move is the same record - the question is how often that happens on the bench workload.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so101_sim_amd.model import scenes
from tests.simharness import ArraySim
raw32, meta = scenes.load_blob("banana", "f32")
N = 4096
s = ArraySim(raw32, N, backend="gpu", seed=0, last_step=100000, prefetch_resets=0)
s.reset()
lo = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0], dtype=np.float32)
hi = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08], dtype=np.float32)
rng = np.random.RandomState(2)
prev = s.qpos.clone()
rows = []
for t in range(int(os.environ.get("STEPS", "300"))):
    s.step(rng.uniform(lo, hi, size=(N, 6)).astype(np.float32))
    q = s.qpos
    same_obj = (q[6:13] == prev[6:13]).all(0); same_con = (q[13:20] == prev[13:20]).all(0)
    v = s.qvel
    rows.append((t, float(same_obj.float().mean()), float(same_con.float().mean()), float((same_obj & same_con).float().mean()),
                 float(v[6:12].abs().max(0).values.median()), float(v[12:18].abs().max(0).values.median())))
    prev = q.clone()
for r in rows:
    if r[0] < 10 or r[0] % 10 == 0:
        print("step %3d: object pose unchanged over the control step in %.3f of the envs, container %.3f, both %.3f; median max|qvel| object %.2e container %.2e" % r)
# substep resolution on the last state
prev = s.qpos.clone()
for k in range(10):
    s.sim.physics(1, 0)
    q = s.qpos
    print("substep %d: object unchanged %.3f container %.3f" % (k, float((q[6:13] == prev[6:13]).all(0).float().mean()), float((q[13:20] == prev[13:20]).all(0).float().mean())))
    prev = q.clone()
