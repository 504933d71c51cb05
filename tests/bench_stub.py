"""Stub env factory for tests/test_bench_launcher.py: stands in for task_suite.create_task_env so that the REAL bench.py
main (rank spawner, sharding by global env id, barrier-bracketed windows, MAX-over-ranks time, rank-ordered all-gather of
returns) runs on CPU with gloo.  The higher rank steps slower, so the job time must be its time."""
import os
import time

import numpy as np
import torch


class _Spec:
    minimum = np.array([-np.pi, -3.14158, -3.14158, -3.14158, -3.14158, 0.0], np.float32)
    maximum = np.array([np.pi, 3.14158, 3.14158, 3.14158, 3.14158, 0.08], np.float32)


class _Sim:
    def configure(self, **kw):
        pass


class StubEnv:
    def __init__(self, name, n_envs, env_id_base, device, **kw):
        self.n_envs, self.sim = n_envs, _Sim()
        self.ids = torch.arange(env_id_base, env_id_base + n_envs, dtype=torch.float32)
        self.ret = torch.zeros(n_envs)
        self.reward = torch.zeros(n_envs)
        self.obs = torch.zeros(n_envs, 18)
        self.step_type = torch.ones(n_envs, dtype=torch.uint8)
        self.delay = 0.01 * (1 + 2 * int(os.environ.get("RANK", "0")))

    def action_spec(self):
        return _Spec()

    def reset_all(self):
        self.ret.zero_()

    def step_tensor(self, act):
        assert act.shape == (self.n_envs, 6)
        time.sleep(self.delay)
        self.ret += self.ids          # a function of the GLOBAL env id, like the kernels' RNG keying

    def diagnostics(self):
        return torch.zeros(self.n_envs, 8, dtype=torch.int32)

    def events(self, clear=False):
        return {"diverged": 0}

    def episode_returns(self):
        return self.ret.clone()


class _TreeSim:
    nq, nv, build = 30, 28, 32


class TreeStubEnv(StubEnv):
    """stands in for AlohaEnvironment (`bench.py --workload aloha | dining`): 14 action dimensions, step_tensor returns the four outputs"""
    def __init__(self, name, n_envs, env_id_base, device, **kw):
        super().__init__(name, n_envs, env_id_base, device, **kw)
        self.sim = _TreeSim()

    def action_spec(self):
        class S:
            minimum = -np.ones(14, np.float32) * 3.0
            maximum = np.ones(14, np.float32) * 3.0
        return S()

    def reset(self):
        self.ret.zero_()

    def step_tensor(self, act):
        assert act.shape == (self.n_envs, 14)
        time.sleep(self.delay)
        self.ret += self.ids
        return self.obs, self.reward, torch.ones(self.n_envs), self.step_type

    def launch_plan(self):
        return {"path": "stub", "slices": 1, "kernel_launches": 1, "memsets": 0}

    def close(self):
        pass


def make(name, n_envs, env_id_base, device, **kw):
    if kw.get("workload") in ("aloha", "dining"):
        return TreeStubEnv(name, n_envs, env_id_base, device, **kw)
    return StubEnv(name, n_envs, env_id_base, device, **kw)
