"""Wall time per block of BLOCK (default 50) control steps over several episodes (4096 envs, bench workload): shows what the synchronous
mass reset at the time limit and the reset prefetch cost."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so101_sim_amd import task_suite
N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1600
BLOCK = int(os.environ.get("BLOCK", "50"))
env = task_suite.create_task_env("SO100HandOverBanana", time_limit=10.0, n_envs=N, random_state=0, device="cuda:0")
env.reset_all()
lo = torch.tensor([-3.14159, -3.14158, -3.14158, -3.14158, -3.14158, 0.0], device="cuda:0")
hi = torch.tensor([3.14159, 3.14158, 3.14158, 3.14158, 3.14158, 0.08], device="cuda:0")
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
stream = torch.cuda.Stream(device="cuda:0") if len(sys.argv) > 3 else torch.cuda.current_stream()   # (a side stream, as bench.py uses)
torch.cuda.synchronize(); t0 = time.perf_counter()
torch.cuda.set_stream(stream)
for i in range(steps):
    a = lo + (hi - lo) * torch.rand(N, 6, device="cuda:0", generator=g)
    env.step_tensor(a)
    if (i + 1) % BLOCK == 0:
        stream.synchronize(); t1 = time.perf_counter()      # (the stepping stream only: a device-wide sync would also wait for the background prefetch)
        d = env.diagnostics().float().mean(0)
        print("steps %4d-%4d: %6.2f ms/step  contacts %.1f iters %.2f  episode min/max %d/%d  events %s" % (
            i - BLOCK + 2, i + 1, (t1 - t0) * 1e3 / BLOCK, d[0], d[2], int(env.episode.min()), int(env.episode.max()),
            {k: v for k, v in env.events().items() if v}))
        stream.synchronize(); t0 = time.perf_counter()
