"""Dining scene: contacts / rows / flags of every env right after the in-call reset (placement + settle), and after a few steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from so101_sim_amd.model import scenes
from tests.simharness import TreeArraySim
task = sys.argv[1] if len(sys.argv) > 1 else "banana"
n = 16
raw32 = (scenes.load_dining_blob(task, "f32")[0] if task in ("banana", "mug", "pen") else scenes.load_aloha_blob("banana", "f32")[0])
sim = TreeArraySim(raw32, n, backend="gpu")
sim.enable_env(seed=11, env_id_base=2, last_step=50, reward_mode=0)
sim.step(np.zeros((n, 14)))
d = sim.get_diag()
print(task, "after reset: contacts", d[:, 0], "rows", d[:, 1], "flags", d[:, 4])
for k in range(3):
    sim.step(np.zeros((n, 14)))
    d = sim.get_diag()
    print(" step", k, "contacts", d[:, 0].max(), "rows", d[:, 1].max(), "flags", np.bitwise_or.reduce(d[:, 4]))
