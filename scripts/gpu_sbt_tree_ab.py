"""Support-bound tables in the general-tree engine's broadphase (round 6): bit-identity of rollouts across auto-resets with and without them
(SO101_NO_SBT=1), no NaN after in-call resets, throughput both ways."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "state":
    import numpy as np, torch
    from so101_sim_amd import task_suite
    name, n, steps, out, prefetch = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6])
    os.chdir("/tmp")
    env = task_suite.create_task_env(name, time_limit=0.2, random_state=0, n_envs=n, prefetch_resets=bool(prefetch))
    env.reset()
    gen = torch.Generator(device=env.device); gen.manual_seed(3)
    spec = env.action_spec()
    lo, hi = torch.tensor(spec.minimum, device=env.device), torch.tensor(spec.maximum, device=env.device)
    acc, nc = [], []
    for t in range(steps):
        a = lo + (hi - lo) * torch.rand(n, len(spec.minimum), device=env.device, generator=gen)
        obs, r, d, st = env.step_tensor(a)
        torch.cuda.synchronize()
        acc.append(np.concatenate([env.qpos.cpu().numpy().ravel(), env.qvel.cpu().numpy().ravel(), obs.cpu().numpy().ravel(), r.cpu().numpy(), st.float().cpu().numpy()]))
        dg = env.diagnostics().float(); nc.append((float(dg[:, 3].mean()), float(dg[:, 0].mean())))
    a = np.stack(acc)
    assert np.isfinite(a).all(), "NaN in the rollout"
    np.save(out, a); json.dump(nc, open(out + ".json", "w"))
    sys.exit(0)
import numpy as np
for name, n in (("HandOverBanana", 256), ("DiningPlaceBananaInBowl", 64)):
    for prefetch in (0, 1):
        outs, ncs = [], []
        for off in (0, 1):
            env = dict(os.environ); env.pop("SO101_NO_SBT", None)
            if off: env["SO101_NO_SBT"] = "1"
            f = "/tmp/sbtt_%s_%d_%d.npy" % (name, prefetch, off)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "state", name, str(n), "24", f, str(prefetch)], env=env)
            outs.append(np.load(f)); ncs.append(json.load(open(f + ".json")))
        print(name, "prefetch", prefetch, ": 24 steps of", n, "envs across two time limits, finite, with / without the tables bit-identical:", bool(np.array_equal(outs[0], outs[1])),
              "| candidates per env (step 5) with %.1f without %.1f, contacts %.1f" % (ncs[0][5][0], ncs[1][5][0], ncs[0][5][1]))
for w in ("aloha", "dining"):
    for off in (0, 1, 0, 1):
        env = dict(os.environ); env.pop("SO101_NO_SBT", None)
        if off: env["SO101_NO_SBT"] = "1"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", w, "--steps", "20", "--warmup", "5", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print("%-7s tables %-3s  %8.1f k env-steps/s  %.3f ms/step" % (w, "off" if off else "on", d["value"] / 1e3, d["ms_per_step"]))
