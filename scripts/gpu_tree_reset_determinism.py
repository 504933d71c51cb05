"""Run-to-run determinism of the ALOHA hand-over env across an auto-reset (round 6 finding): the same seed, the same actions, twice per setting."""
import os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "state":
    import numpy as np, torch
    from so101_sim_amd import task_suite
    name, n, steps, out, prefetch, pipeline = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
    os.chdir("/tmp")
    env = task_suite.create_task_env(name, time_limit=0.2, random_state=0, n_envs=n, prefetch_resets=bool(prefetch), pipeline=bool(pipeline))
    env.reset()
    gen = torch.Generator(device=env.device); gen.manual_seed(3)
    spec = env.action_spec()
    lo, hi = torch.tensor(spec.minimum, device=env.device), torch.tensor(spec.maximum, device=env.device)
    acc = []
    for t in range(steps):
        a = lo + (hi - lo) * torch.rand(n, len(spec.minimum), device=env.device, generator=gen)
        obs, r, d, st = env.step_tensor(a)
        torch.cuda.synchronize()
        acc.append(np.concatenate([env.qpos.cpu().numpy().T, env.qvel.cpu().numpy().T, st.float().cpu().numpy()[:, None], env.diagnostics().cpu().numpy().astype(np.float32)], axis=1))
    np.save(out, np.stack(acc))
    sys.exit(0)
import numpy as np
name = sys.argv[1] if len(sys.argv) > 1 else "HandOverBanana"
for prefetch, pipeline in ((1, 1), (0, 1), (0, 0)):
    outs = []
    for i in range(2):
        f = "/tmp/det_%d%d_%d.npy" % (prefetch, pipeline, i)
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "state", name, "64", "14", f, str(prefetch), str(pipeline)])
        outs.append(np.load(f))
    a, b = outs
    neq = (a != b).any(axis=2)          # [step][env]
    print(name, "prefetch", prefetch, "pipeline", pipeline, ": differing envs per step", neq.sum(axis=1).tolist())
    if neq.any():
        t, e = np.argwhere(neq)[0]
        print("   first difference: step", t, "env", e, "step_type", a[t, e, -9], b[t, e, -9], "diag", a[t, e, -8:], b[t, e, -8:])
        d = np.abs(a[t, e] - b[t, e]); print("   max |diff|", d.max(), "at column", int(d.argmax()), "of", a.shape[2])
