"""Why do 20-step windows separated by synchronisation slow down?  per-20-step device time, continuous vs windowed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so101_sim_amd import task_suite
dev = torch.device("cuda", 0)
os.chdir("/tmp")
mode = sys.argv[1]
env = task_suite.create_task_env("SO100HandOverBanana", time_limit=10.0, random_state=0, device=dev, n_envs=4096)
st = torch.cuda.Stream(dev)
spec = env.action_spec()
lo = torch.tensor(spec.minimum, device=dev); hi = torch.tensor(spec.maximum, device=dev)
g = torch.Generator(device=dev); g.manual_seed(1)
tape = lo + (hi - lo) * torch.rand(200, 4096, 6, device=dev, generator=g)
with torch.cuda.stream(st):
    env.reset_all()
torch.cuda.synchronize()
ticks = []
with torch.cuda.stream(st):
    for i in range(145):
        if i >= 5 and (i - 5) % 20 == 0:
            if mode == "sync":
                torch.cuda.synchronize()
            if mode == "sleep":
                torch.cuda.synchronize(); time.sleep(0.05)
            if mode == "events":
                torch.cuda.synchronize(); print(env.events())
            e = torch.cuda.Event(enable_timing=True); e.record(st); ticks.append(e)
        env.step_tensor(tape[i])
torch.cuda.synchronize()
print(mode, [round(4096 * 20 / (ticks[k].elapsed_time(ticks[k + 1]) * 1e-3)) for k in range(len(ticks) - 1)], env.events())
