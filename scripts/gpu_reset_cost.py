"""Cost of env.reset() for 4096 envs: placement + settle every time, against the settled-state store (SURVEY 8f-3:
states computed once, kept on disk, copied at reset)."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from so101_sim_amd import task_suite

N = 4096
mk = lambda: task_suite.create_task_env("SO100HandOverBanana", time_limit=10.0, n_envs=N, random_state=0, device="cuda:0",
                                        prefetch_resets=False)


def timed(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


env = mk()
t_settle = timed(env.reset_all)
print("reset of %d envs, placement + settle (<= 1000 substeps each): %.1f ms; events %s" % (N, t_settle, env.events()))
path = os.path.join(tempfile.mkdtemp(), "settled.bin")
t_build = timed(lambda: env.save_settled_cache(path, n_episodes=4))
print("settled-state file for 4 episodes per env: computed + written in %.1f ms, %.1f MB" % (t_build, os.path.getsize(path) / 1e6))
env.close()
env = mk()
t_load = timed(lambda: env.load_settled_cache(path))
t_copy = timed(env.reset_all)
print("second process: file loaded + attached in %.1f ms; reset of %d envs from the store: %.2f ms (%.0fx)" % (t_load, N, t_copy, t_settle / t_copy))
