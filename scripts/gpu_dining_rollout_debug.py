"""tests/test_dining.py::test_rollout_against_the_oracle with its numbers printed per env (props: position, linear and angular velocity differences)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from so101_sim_amd.model import scenes
from tests.simharness import TreeArraySim
from oracle.oracle import Oracle
import tests.test_dining as TD
n, steps = 8, 5
raw64, raw32 = scenes.load_dining_blob("banana", "f64")[0], scenes.load_dining_blob("banana", "f32")[0]
sim = TreeArraySim(raw32, n, backend="gpu")
Q, V, W, CT = TD._dining_states(n, 1)
sim.set_state(Q, V, CT, W)
for _ in range(steps):
    sim.physics(10)
q1, v1, _ = sim.get_state()
print("flags", sim.get_diag()[:, 4], "contacts", sim.get_diag()[:, 0])
multi = int(os.environ.get("ORACLE_MULTI", "1"))
o = Oracle(raw64); o.set_hull_multicontact(bool(multi))
for e in range(n):
    o.set_state(Q[:, e], V[:, e], W[:, e]); o.set_ctrl(CT[:, e])
    for _ in range(steps):
        o.substeps(10, False)
    q, v, _ = o.get_state()
    pv = (v1[16:, e] - v[16:]).reshape(6, 6)
    dq = np.abs(q1[16:, e] - q[16:]).reshape(6, 7)
    print("env", e, "arm dq %.1e" % np.abs(q1[:16, e] - q[:16]).max(), "props dq per prop", np.round(dq.max(1) * 1e4, 2), "e-4  lin vel %.1e ang vel %.2e" % (np.abs(pv[:, :3]).max(), np.abs(pv[:, 3:]).max()),
          "oracle contacts", len(o.contacts()))
# contact lists of env 0 at the kernel's final state
from tests import parity_cases as pc
meta = scenes.load_dining_blob("banana", "f64")[1]; gn = meta["geom_names"]
sim2 = TreeArraySim(raw32, 1, backend="gpu")
_, _, w1 = sim.get_state()
sim2.set_state(q1[:, :1], v1[:, :1], CT[:, :1], w1[:, :1])
d = sim2.debug_forward()[0]
o.set_state(q1[:, 0].astype(np.float64), v1[:, 0].astype(np.float64), w1[:, 0].astype(np.float64)); o.set_ctrl(CT[:, 0]); o.forward()
ref = o.contacts()
problems, tot, loose, wit = pc._compare_contact_lists(d["contacts"], ref)
print("env 0 final state: kernel", len(d["contacts"]), "oracle", len(ref), "problems", [(gn[int(p.split(',')[0][1:])], gn[int(p.split(',')[1].split(')')[0])], p.split(':')[1]) for p in problems])
import collections
per = collections.Counter((gn[c["geom1"]], gn[c["geom2"]]) for c in ref)
print({k: v for k, v in per.items() if "table" not in k[0] and "table" not in k[1]})
for c in ref:
    if (gn[c["geom1"]], gn[c["geom2"]]) in [tuple(x[:2]) for x in [(gn[int(p.split(',')[0][1:])], gn[int(p.split(',')[1].split(')')[0])]) for p in problems]]:
        print("  oracle", gn[c["geom1"]], gn[c["geom2"]], "dist %.6f" % c["dist"], np.round(c["pos"], 4), np.round(c["normal"], 4))
for c in d["contacts"]:
    if (gn[c["geom1"]], gn[c["geom2"]]) in [tuple(x[:2]) for x in [(gn[int(p.split(',')[0][1:])], gn[int(p.split(',')[1].split(')')[0])]) for p in problems]]:
        print("  kernel", gn[c["geom1"]], gn[c["geom2"]], "dist %.6f" % c["dist"], np.round(c["pos"], 4), np.round(c["normal"], 4))
