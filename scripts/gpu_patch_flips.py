"""Where do kernel (fp32) and oracle (fp64) contact lists differ since hull pairs carry patches?  A 4096-env random-action rollout supplies
states; a sample of envs is compared contact by contact (tests/parity_cases._compare_contact_lists) at the end of the rollout."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from so101_sim_amd.model import scenes
from tests.simharness import ArraySim
from tests import parity_cases as pc
from oracle.oracle import Oracle
raw32, meta = scenes.load_blob("banana", "f32"); raw64, _ = scenes.load_blob("banana", "f64")
gn = meta["geom_names"]
N, steps = 2048, int(sys.argv[1]) if len(sys.argv) > 1 else 40
s = ArraySim(raw32, N, backend="gpu", seed=3, settle_max_substeps=300, last_step=100000, prefetch_resets=0)
s.reset()
lo = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0], dtype=np.float32); hi = -lo; hi[5] = 0.08
rng = np.random.RandomState(2)
for t in range(steps):
    s.step(rng.uniform(lo, hi, size=(N, 6)).astype(np.float32))
q, v, w = s.get_state(); c = s._get(s.ctrl)
dbg = s.debug_forward()
bad, total, npatch = 0, 0, 0
o = Oracle(raw64)
kinds = collections.Counter()
shown = 0
for e in range(0, N, 4):
    d = dbg[e]
    if d["overflow"]:
        continue
    o.set_state(q[:, e], v[:, e], w[:, e]); o.set_ctrl(c[:, e].astype(np.float64)); o.forward()
    ref = o.contacts()
    problems, tot, loose, wit = pc._compare_contact_lists(d["contacts"], ref)
    total += 1
    per = collections.Counter((cc["geom1"], cc["geom2"]) for cc in ref)
    npatch += sum(1 for k, n in per.items() if n > 1 and gn[k[0]] not in ("table_surface", "floor") and "pad" not in gn[k[0]] and "pad" not in gn[k[1]] and gn[k[1]] != "table_surface")
    if problems:
        bad += 1
        for p in problems:
            kinds[p.split(":")[1].strip().split(" ")[0]] += 1
        if shown < 12:
            shown += 1
            print("env", e, [(gn[int(p.split(",")[0][1:])], gn[int(p.split(",")[1].split(")")[0])], p.split(":")[1]) for p in problems][:4])
print("states compared", total, "with problems", bad, "kinds", dict(kinds), "hull pairs with more than one contact (oracle)", npatch)
