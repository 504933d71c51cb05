"""One-step parity outliers of tests/test_gpu_workloads.py::test_failure_rates_on_the_headline_workload, taken apart: the same rollout
(seed from argv), the probes whose one-step error against the fp64 oracle exceeds 2e-3 rad / 0.1 rad/s, and for each of them the ten
substeps side by side - GPU (so101_physics on a one-env handle) against the oracle from the SAME start state - with the contact lists of
both at the first substep where they part."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so101_sim_amd.model import scenes
from oracle.oracle import Oracle
from tests.simharness import ArraySim
from tests import parity_cases as pc
from tests.test_gpu_workloads import _batched_env

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n, steps, k = 4096, 500, int(os.environ.get("PROBE_ENVS", "24"))
env = _batched_env("SO100HandOverBanana", n)
spec = env.action_spec()
lo = torch.tensor(spec.minimum, device=env.device); hi = torch.tensor(spec.maximum, device=env.device)
gen = torch.Generator(device=env.device); gen.manual_seed(seed)
st = torch.cuda.Stream()
raw64, meta = scenes.load_blob("banana", "f64")
raw32, _ = scenes.load_blob("banana", "f32")
gn = meta["geom_names"]
PROBE_STEPS = tuple(int(x) for x in os.environ.get("PROBE_STEPS", "60,180,300,420,480").split(","))
checks = []
with torch.cuda.stream(st):
    env.reset_all()
    for t in range(steps):
        act = lo + (hi - lo) * torch.rand(n, 6, device=env.device, generator=gen)
        probe = t in PROBE_STEPS
        if probe:
            before = [x[:, :k].clone() for x in (env.qpos, env.qvel, env.warm)]
        env.step_tensor(act)
        if probe:
            checks.append((t, before, act[:k].clone(), env.qpos[:, :k].clone(), env.qvel[:, :k].clone(), env.step_type[:k].clone()))
torch.cuda.synchronize()
bad = []
for t, before, act, q1, v1, stp in checks:
    b = [x.cpu().numpy().astype(np.float64) for x in before]
    for e in range(k):
        if int(stp[e]) != 1:
            continue
        o = Oracle(raw64)
        o.set_state(b[0][:, e], b[1][:, e], b[2][:, e]); o.set_ctrl(act[e].cpu().numpy().astype(np.float64)); o.substeps(10)
        qo, vo, _ = o.get_state()
        dq, dv = np.abs(q1[:, e].cpu().numpy() - qo).max(), np.abs(v1[:, e].cpu().numpy() - vo).max()
        if dq > 2e-3 or dv > 0.1:
            bad.append((t, e, dq, dv, b[0][:, e], b[1][:, e], b[2][:, e], act[e].cpu().numpy().astype(np.float64)))
print("seed %d: %d probes outside 2e-3 rad / 0.1 rad/s" % (seed, len(bad)))
names = lambda c: "%s|%s" % (gn[c["geom1"]], gn[c["geom2"]])
saved = []
for t, e, dq, dv, q, v, w, a in bad:
    print("== step %d env %d: one control step differs by %.3e rad / %.3e rad/s" % (t, e, dq, dv))
    sim = ArraySim(raw32, 1, backend="gpu", seed=0, prefetch_resets=0)
    o = Oracle(raw64)
    o.set_state(q, v, w); o.set_ctrl(a)
    sim.set_state(q[:, None], v[:, None], a[:, None], w[:, None])
    saved.append(dict(step=t, env=e, qpos=q.tolist(), qvel=v.tolist(), warm=w.tolist(), action=a.tolist(), dq=float(dq), dv=float(dv)))
    shown = False
    for s_ in range(10):
        d = sim.debug_forward()[0]
        o.forward()
        ref = o.contacts()
        problems, total, loose, witness = pc._compare_contact_lists(d["contacts"], ref)
        acc = o.qacc()[0]
        da = np.abs(d["qacc"] - acc).max() / max(np.abs(acc).max(), 1e-9)
        sim.physics(1); o.substeps(1)
        qg, vg, _ = sim.get_state(); qo, vo, _ = o.get_state()
        print("  substep %d: contacts gpu %d oracle %d, list problems %d (loose %d, witness %d), qacc rel diff %.2e | after: dq %.2e dv %.2e (dof %d)" % (
            s_, len(d["contacts"]), len(ref), len(problems), loose, witness, da, np.abs(qg[:, 0] - qo).max(), np.abs(vg[:, 0] - vo).max(), int(np.abs(vg[:, 0] - vo).argmax())))
        if (problems or da > 1e-3) and not shown:
            shown = True
            for p in problems[:8]:
                print("      ", p)
            print("      gpu   :", [(names(c), round(c["dist"] * 1e3, 4)) for c in d["contacts"]])
            print("      oracle:", [(names(c), round(c["dist"] * 1e3, 4)) for c in ref])
        # continue both from the GPU's state so that every substep is a one-substep comparison
        o.set_state(qg[:, 0].astype(np.float64), vg[:, 0].astype(np.float64), sim.get_state()[2][:, 0].astype(np.float64))
import json
os.makedirs("gpurun_out", exist_ok=True)
json.dump(saved, open("gpurun_out/probe_outliers_seed%d.json" % seed, "w"))
