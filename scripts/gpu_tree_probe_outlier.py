"""Differential rollout of the general-tree engine (ALOHA hand-over / Dining scenes) against the fp64 oracle: N envs from oracle resets,
random joint targets around the home pose redrawn every control step (the bench workload's distribution), and EVERY control step of every
env repeated by the oracle from the kernel's own start state.  Prints the distribution of the one-step differences and takes the worst
ones apart substep by substep (contact lists of both at the first substep where they part).
    python scripts/gpu_tree_probe_outlier.py [banana|pen|dining] [seed] [n_envs] [steps] [amplitude]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from so101_sim_amd.model import scenes
from oracle.oracle import Oracle
from tests.simharness import TreeArraySim
from tests import parity_cases as pc

name = sys.argv[1] if len(sys.argv) > 1 else "banana"
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n = int(sys.argv[3]) if len(sys.argv) > 3 else 48
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
amp = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5      # width of the uniform joint-target noise around the home pose (the bench workload: 0.5)
if name == "dining":
    raw64, meta = scenes.load_dining_blob("banana", "f64"); raw32, _ = scenes.load_dining_blob("banana", "f32")
else:
    raw64, meta = scenes.load_aloha_blob(name, "f64"); raw32, _ = scenes.load_aloha_blob(name, "f32")
gn = meta["geom_names"]
o = Oracle(raw64)
rng = np.random.RandomState(seed)
Q, V, W = [], [], []
for e in range(n):
    o.env_config(seed=seed, env_id=e); o.env_reset()
    q, v, w = o.get_state(); Q.append(q); V.append(v); W.append(w)
Q, V, W = (np.array(x).T for x in (Q, V, W))
home = np.concatenate([scenes.ALOHA_HOME_CTRL] * 2)
sim = TreeArraySim(raw32, n, backend="gpu")
sim.set_state(Q, V, np.tile(home[:, None], (1, n)), W)
errs, worst = [], []
for t in range(steps):
    ctrl = home[:, None] + amp * (rng.rand(14, n) - 0.5)
    ctrl[6], ctrl[13] = rng.uniform(0.002, 0.037, size=n), rng.uniform(0.002, 0.037, size=n)
    q0, v0, w0 = sim.get_state()
    sim.set_state(ctrl=ctrl)
    sim.physics(10)
    q1, v1, _ = sim.get_state()
    flags = sim.get_diag()[:, 4]
    for e in range(n):
        if flags[e] & 8:
            continue
        o.inject_contacts([]); o.set_state(q0[:, e], v0[:, e], w0[:, e]); o.set_ctrl(ctrl[:, e]); o.substeps(10)
        qo, vo, _ = o.get_state()
        dq, dv = np.abs(q1[:, e] - qo).max(), np.abs(v1[:, e] - vo).max()
        errs.append((dq, dv))
        if dq > 2e-3 or dv > 0.1:
            worst.append((dq, dv, t, e, q0[:, e].copy(), v0[:, e].copy(), w0[:, e].copy(), ctrl[:, e].copy()))
errs = np.array(errs)
print("amplitude %.1f: " % amp, end="")
print("%s seed %d: %d one-step comparisons; dq median %.2e p99 %.2e max %.2e | dv median %.2e p99 %.2e max %.2e | outside 2e-3 / 0.1: %d (%.2f %%)" % (
    name, seed, len(errs), np.median(errs[:, 0]), np.percentile(errs[:, 0], 99), errs[:, 0].max(), np.median(errs[:, 1]), np.percentile(errs[:, 1], 99), errs[:, 1].max(),
    len(worst), 100.0 * len(worst) / len(errs)))
names = lambda c: "%s|%s" % (gn[c["geom1"]], gn[c["geom2"]])
saved = []
one = TreeArraySim(raw32, 1, backend="gpu")
for dq, dv, t, e, q, v, w, c in sorted(worst, key=lambda x: -x[1])[:12]:
    print("== step %d env %d: %.3e rad|m / %.3e" % (t, e, dq, dv))
    saved.append(dict(step=t, env=e, qpos=q.tolist(), qvel=v.tolist(), warm=w.tolist(), ctrl=c.tolist(), dq=float(dq), dv=float(dv)))
    one.set_state(q[:, None], v[:, None], c[:, None], w[:, None])
    o.inject_contacts([]); o.set_state(q, v, w); o.set_ctrl(c)
    shown = False
    for s_ in range(10):
        d = one.debug_forward()[0]
        o.forward()
        ref = o.contacts()
        problems, total, loose, witness = pc._compare_contact_lists(d["contacts"], ref)
        acc = o.qacc()[0]
        da = np.abs(d["qacc"] - acc).max() / max(np.abs(acc).max(), 1e-9)
        one.physics(1); o.substeps(1)
        qg, vg, wg = one.get_state(); qo, vo, _ = o.get_state()
        print("  substep %d: contacts gpu %d oracle %d, list problems %d (loose %d, witness %d), qacc rel diff %.2e | after: dq %.2e dv %.2e (dof %d)" % (
            s_, len(d["contacts"]), len(ref), len(problems), loose, witness, da, np.abs(qg[:, 0] - qo).max(), np.abs(vg[:, 0] - vo).max(), int(np.abs(vg[:, 0] - vo).argmax())))
        if (problems or da > 1e-2) and not shown:
            shown = True
            for p in problems[:8]:
                print("      ", p)
            print("      gpu   :", [(names(c_), round(c_["dist"] * 1e3, 4)) for c_ in d["contacts"]])
            print("      oracle:", [(names(c_), round(c_["dist"] * 1e3, 4)) for c_ in ref])
        o.set_state(qg[:, 0], vg[:, 0], wg[:, 0])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(saved, open("gpurun_out/tree_probe_outliers_%s_seed%d.json" % (name, seed), "w"))
