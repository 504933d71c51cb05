"""Where a substep of the general-tree engine spends its time (run through gpurun): so101_tree_debug_forward with the stage mask
SO101_TREE_PHASES (one process per mask: the library reads it once), 2048 envs of the banana scene after reset.
    python scripts/gpu_tree_phases.py [dining]      (dining: 1024 envs of DiningPlaceBananaInBowl, the 64-dof build)"""
import os
import subprocess
import sys
import time

if len(sys.argv) > 1 and "SO101_TREE_PHASES" in os.environ:
    import numpy as np
    import torch
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    from so101_sim_amd import task_suite
    dining = sys.argv[1] == "dining"
    n = 1024 if dining else 2048
    env = task_suite.create_task_env("DiningPlaceBananaInBowl" if dining else "HandOverBanana", time_limit=10.0, random_state=0, n_envs=n, settle_max_substeps=200)
    env.reset()
    dbg = torch.zeros(n, env.sim.debug_dim, device=env.device)
    for _ in range(2):
        env.sim.debug_forward(dbg.data_ptr(), 0)
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(5):
        env.sim.debug_forward(dbg.data_ptr(), 0)
    torch.cuda.synchronize()
    print("mask %3d: %.3f ms per forward of %d envs" % (int(os.environ["SO101_TREE_PHASES"]), (time.time() - t0) / 5 * 1e3, n), flush=True)
else:
    for mask in (1, 3, 7, 15, 31 + 128, 31, 63, 127):        # (+128: the collision stage without its narrowphase)
        subprocess.run([sys.executable, __file__, "dining" if "dining" in sys.argv[1:] else "x"], env=dict(os.environ, SO101_TREE_PHASES=str(mask)))
