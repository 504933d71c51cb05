"""Workload for rocprofv3 --pmc on the stepping kernels: 4096 envs, random actions, a few control steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from so101_sim_amd.model import scenes
from tests.simharness import ArraySim
raw32, _ = scenes.load_blob("banana", "f32")
N = 4096
s = ArraySim(raw32, N, backend="gpu", seed=0, settle_max_substeps=100, last_step=100000, prefetch_resets=0,
             pipeline=int(os.environ.get("PIPELINE", "1")))
s.reset()
lo = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0], dtype=np.float32)
hi = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08], dtype=np.float32)
rng = np.random.RandomState(2)
for t in range(int(os.environ.get("STEPS", "14"))):
    s.step(rng.uniform(lo, hi, size=(N, 6)).astype(np.float32))
torch.cuda.synchronize()
print("done")
