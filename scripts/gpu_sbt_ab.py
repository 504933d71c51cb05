"""Support-bound tables in the broadphase / support-vertex lists in the narrowphase (round 6): bit-identity of a 4096-env rollout with and without
them (SO101_NO_SBT=1 / SO101_NO_HL=1 upload no tables at so101_create; argv[1] of the top-level call names the variable, default SO101_NO_SBT),
the candidate counts, and the throughput of the driver's command both ways."""
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "state":
    import numpy as np, torch
    from tests.test_gpu_workloads import _batched_env
    n, steps = 4096, int(sys.argv[2])
    env = _batched_env("SO100HandOverBanana", n, time_limit=1.0)
    spec = env.action_spec()
    lo = torch.tensor(spec.minimum, device=env.device); hi = torch.tensor(spec.maximum, device=env.device)
    gen = torch.Generator(device=env.device); gen.manual_seed(11)
    st = torch.cuda.Stream()
    acc, ncand = [], []
    with torch.cuda.stream(st):
        env.reset_all()
        for t in range(steps):
            env.step_tensor(lo + (hi - lo) * torch.rand(n, 6, device=env.device, generator=gen))
            if t % 10 == 9:
                acc.append(torch.cat([env.qpos.flatten(), env.qvel.flatten(), env.reward, env.discount, env.step_type.float()]).cpu().numpy())
                d = env.diagnostics().float(); ncand.append((float(d[:, 3].mean()), float(d[:, 0].mean())))
    torch.cuda.synchronize()
    assert all(np.isfinite(a).all() for a in acc)
    np.save(sys.argv[3], np.concatenate(acc)); json.dump(ncand, open(sys.argv[3] + ".json", "w"))
    sys.exit(0)
import numpy as np
VAR = sys.argv[1] if len(sys.argv) > 1 else "SO101_NO_SBT"
outs, nc = [], []
for off in (0, 1):
    env = dict(os.environ); env.pop(VAR, None)
    if off: env[VAR] = "1"
    f = "/tmp/sbt_%d.npy" % off
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "state", "120", f], env=env)
    outs.append(np.load(f)); nc.append(json.load(open(f + ".json")))
print("120 control steps of 4096 envs across two time limits (random actions, resets with the prefetch): with and without the tables bit-identical:", bool(np.array_equal(outs[0], outs[1])), "(", outs[0].size, "numbers, all finite )")
print("mean narrowphase candidates / contacts per env at steps 10, 60, 120:  with", [tuple(round(x, 2) for x in nc[0][i]) for i in (0, 5, 11)], " without", [tuple(round(x, 2) for x in nc[1][i]) for i in (0, 5, 11)])
for off in (0, 1, 0, 1):
    env = dict(os.environ); env.pop(VAR, None)
    if off: env[VAR] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no-cpu-baseline"], env=env, capture_output=True, text=True)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    print(VAR + " %-3s  %8.1f k env-steps/s  windows %s  candidates per env %.2f" % ("set" if off else "-", d["value"] / 1e3, [round(v / 1e3) for v in d["repeats"]["values"]], d["diag_mean"]["narrowphase_candidates"]))
