"""Times so101_physics on a settled 4096-env batch with stages masked off (SO101_DEBUG_PHASES) to attribute cost."""
import os, sys, time, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import numpy as np, torch
    from so101_sim_amd.model import scenes
    from tests.simharness import ArraySim
    raw32, _ = scenes.load_blob("banana", "f32")
    N, iters = 4096, int(sys.argv[2])
    s = ArraySim(raw32, N, backend="gpu", seed=0, solver_iterations=iters, settle_max_substeps=300)
    os.environ.pop("SO101_DEBUG_PHASES", None)
    ph = os.environ.get("PHASES", "7")
    s.reset(); torch.cuda.synchronize()
    os.environ["SO101_DEBUG_PHASES"] = ph
    s.physics(10); torch.cuda.synchronize()
    t = time.time()
    for _ in range(5): s.physics(10)
    torch.cuda.synchronize(); dt = (time.time() - t) / 5
    print(json.dumps(dict(phases=int(ph), iters=iters, ms_per_control_step=dt * 1e3, diag=s.get_diag()[:1].tolist())))
else:
    for iters in (100, 10):
        for ph in ("0", "1", "3", "7"):
            env = dict(os.environ, PHASES=ph)
            print(subprocess.run([sys.executable, __file__, "child", str(iters)], env=env, capture_output=True, text=True).stdout.strip())
