"""Captures contact-rich states from a random-action rollout of the CURRENT build (default narrowphase: EPA) on the GPU and appends
them to tests/golden/contact_rich_states.json (inputs only: qpos, qvel, warm start, ctrl; the expected values come from the oracle
at test time).  Selection: finite, |qvel| < 60, at least 10 contacts, preferring states with arm-arm, arm-prop and arm-table contacts.
    usage (through gpurun): python scripts/gpu_capture_contact_states.py OUT.json [count]"""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.simharness import ArraySim
from so101_sim_amd.model import scenes

out_path = sys.argv[1]; count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
raw32, _ = scenes.load_blob("banana", "f32")
n = 1024
sim = ArraySim(raw32, n, backend="gpu", seed=11, last_step=500, settle_max_substeps=300)
sim.reset()
rng = np.random.RandomState(5)
lo = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0]); hi = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08])
arm = lambda g: 1 <= g <= 18
picked = []
for t in range(160):
    act = rng.uniform(lo, hi, size=(n, 6)).astype(np.float32)
    sim.step(act)
    if t < 30 or t % 10:
        continue
    q, v, w = sim.get_state()
    c = sim._get(sim.ctrl).astype(np.float64)
    dbg = sim.debug_forward()
    for e in range(n):
        d = dbg[e]
        if d["overflow"] or d["ncon"] < 10 or not np.all(np.isfinite(q[:, e])) or np.abs(v[:, e]).max() > 60:
            continue
        pairs = [(k["geom1"], k["geom2"]) for k in d["contacts"]]
        aa = sum(arm(a) and arm(b) for a, b in pairs)
        ap = sum((arm(a) and b >= 26) or (arm(b) and a >= 26) for a, b in pairs)
        at = sum((arm(a) and 19 <= b <= 25) or (arm(b) and 19 <= a <= 25) for a, b in pairs)
        deep = sum(k["dist"] < -2e-3 for k in d["contacts"])
        score = 3 * min(aa, 3) + 2 * min(ap, 4) + min(at, 3) + min(deep, 3)
        if aa + ap + at >= 3:
            picked.append((score, t, e, dict(qpos=q[:, e].tolist(), qvel=v[:, e].tolist(), warm=w[:, e].tolist(), action=c[:, e].tolist()),
                           dict(ncon=int(d["ncon"]), arm_arm=int(aa), arm_prop=int(ap), arm_table=int(at), deep=int(deep))))
picked.sort(key=lambda x: -x[0])
chosen, seen = [], set()
for s, t, e, st, info in picked:
    if e in seen:
        continue
    seen.add(e); chosen.append((st, info, t, e))
    if len(chosen) == count:
        break
json.dump({"states": [c[0] for c in chosen], "info": [dict(c[1], step=int(c[2]), env=int(c[3])) for c in chosen]}, open(out_path, "w"))
for c in chosen:
    print(c[1], "step", c[2], "env", c[3])
