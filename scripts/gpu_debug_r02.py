import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so101_sim_amd import task_suite, pregrasp
from so101_sim_amd.model import scenes
from oracle.oracle import Oracle
from tests.simharness import ArraySim
raw32, _ = scenes.load_blob("banana", "f32"); raw64, _ = scenes.load_blob("banana", "f64")
cwd = os.getcwd(); os.chdir("/tmp")
n = 64
env = task_suite.create_task_env("SO100HandOverBanana", time_limit=10.0, random_state=0, n_envs=n)
PQ, PV, PC = (t.cpu().numpy().astype(np.float64) for t in pregrasp.build_pickplace_pool(env, pool_size=n, seed=3))
env.close(); os.chdir(cwd)
np.savez("gpurun_out/pool_debug.npz", PQ=PQ, PV=PV, PC=PC)
idx = list(range(8))
for iters in (100, 300):
    sim = ArraySim(raw32, len(idx), backend="gpu", last_step=500, solver_iterations=iters)
    sim.set_state(PQ[:, idx], PV[:, idx], PC[:, idx], np.zeros((18, len(idx))))
    dbg = sim.debug_forward()
    for j, k in enumerate(idx):
        o = Oracle(raw64); o.set_solver(iters, -1.0); o.set_state(PQ[:, k], PV[:, k], np.zeros(18)); o.set_ctrl(PC[:, k]); o.forward()
        a, asm = o.qacc()
        err = np.abs(dbg[j]["qacc"] - a)
        print("iters cap", iters, "entry", k, "gpu iters", dbg[j]["iters"], "oracle iters", o.solver_iter if hasattr(o, "solver_iter") else "?", "ncon", dbg[j]["ncon"],
              "max|a| %.1f" % np.abs(a).max(), "rel err %.2e" % (err.max() / np.abs(a).max()), "worst dof", int(err.argmax()), "a_o %.3f a_g %.3f" % (a[err.argmax()], dbg[j]["qacc"][err.argmax()]))
