"""Phase clocks of the constrained half of one forward of the general-tree engine (run through gpurun with a -DTREE_PROF variant library:
    python scripts/build_variant.py treeprof --tus tu_tree,tu_tree64 -- -DTREE_PROF
    SO101_HIP_LIB=ab/lib_treeprof.so python scripts/gpu_tree_prof.py [dining]
State: the envs after reset plus STEPS control steps of random joint targets (the bench workload's contact mix)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so101_sim_amd import task_suite
dining = len(sys.argv) > 1 and sys.argv[1] == "dining"
n = 1024 if dining else 4096
env = task_suite.create_task_env("DiningPlaceBananaInBowl" if dining else "HandOverBanana", time_limit=10.0, random_state=0, n_envs=n, settle_max_substeps=200)
env.reset()
spec = env.action_spec()
g = torch.Generator(device=env.device); g.manual_seed(1)
lo = torch.as_tensor(spec.minimum, device=env.device, dtype=torch.float32); hi = torch.as_tensor(spec.maximum, device=env.device, dtype=torch.float32)
for _ in range(int(os.environ.get("STEPS", "10"))):
    a = lo + (hi - lo) * torch.rand(n, lo.numel(), device=env.device, generator=g)
    env.step_tensor(a)
dbg = torch.zeros(n, env.sim.debug_dim, device=env.device)
env.sim.debug_forward(dbg.data_ptr(), 0); torch.cuda.synchronize()
d = dbg.cpu().numpy()
off_m = env.sim.dbg["M"]
print("debug_dim", env.sim.debug_dim, "offset M", off_m)
prof = d[:, off_m:off_m + 16] * 1e-2      # us
ncon, nrow, iters = d[:, 0], d[:, 1], d[:, 2]
names = ["cost: M x + jar", "cost: blocks", "cost: gradient", "cost: H = M", "cost: H scalar rows", "cost: H contacts", "chol factor", "chol solve", "ls setup (J search)", "line search", "make_constraints", "newton total", "", "", "", "loop top (incl. cost call)"]
print("%d envs: contacts %.1f (max %d)  rows %.1f (max %d)  iterations %.2f (max %d)" % (n, ncon.mean(), ncon.max(), nrow.mean(), nrow.max(), iters.mean(), iters.max()))
for k, nm in enumerate(names):
    if nm: print("  %-28s mean %8.1f us   p50 %8.1f   p99 %8.1f   max %8.1f" % (nm, prof[:, k].mean(), np.median(prof[:, k]), np.percentile(prof[:, k], 99), prof[:, k].max()))
env.close()        # (the profiling variant prints the phase clocks of the launch chain's k_tree_pipe_solve when the handle goes)
