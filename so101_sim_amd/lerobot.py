"""LeRobot-format packaging on top of the batched env.

Mirrors `scripts/so101_lerobot_wrapper.py:15-122` of the reference: `reset()` / `step(action)` return a dict with
`observation.state` (the DELAYED `joints_pos`, :101-103), `action` (zeros on reset, :109-112), `timestamp =
frame_index * 0.1` (:115 — 0.1, not the 0.02 s control step: kept), `frame_index`, `episode_index`, `index`,
`task_index`, `task`.  Cameras are outside the MI355X hot path: `observation.images.*` keys are never
produced and `get_observation_spec()['images']` is empty.

Batched extension: `n_envs > 1` keeps every tensor on the GPU with a leading env dimension (`observation.state`
[N, 6], `action` [N, 6], index tensors [N]); with `n_envs == 1` shapes and dtypes are the reference's ((6,), scalar
tensors).  Like the reference wrapper, `frame_index` belongs to the wrapper and only `reset()` rewinds it — the env's
own auto-reset after LAST does not.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import numpy as np

TASK_STRING = "SO100 manipulation task"          # so101_lerobot_wrapper.py:120
JOINT_ACTION_NAMES = ["rotation", "pitch", "elbow", "wrist_pitch", "wrist_roll", "jaw"]


def to_lerobot_format(state, action, frame_index: int, episode_index: int, device=None, n_envs: int = 1):
    """Pure packaging step (so101_lerobot_wrapper.py:77-122).  `state` / `action`: array-likes of shape (6,) or
    (N, 6); `action=None` gives the zero action of the initial observation."""
    import torch
    dev = device if device is not None else "cpu"
    st = torch.as_tensor(state).float().to(dev)
    batched = n_envs > 1
    if action is None:
        act = torch.zeros((n_envs, 6) if batched else (6,)).float().to(dev)
    else:
        act = torch.as_tensor(action).float().to(dev)

    def meta(value, dtype):
        t = torch.tensor(value).to(dtype)
        return (t.expand(n_envs).clone() if batched else t).to(dev)

    return {
        "observation.state": st,
        "action": act,
        "timestamp": meta(frame_index * 0.1, torch.float32),
        "frame_index": meta(frame_index, torch.long),
        "episode_index": meta(episode_index, torch.long),
        "index": meta(frame_index, torch.long),
        "task_index": meta(0, torch.long),
        "task": TASK_STRING,
    }


class SO101LeRobotWrapper:
    def __init__(self, task_name: str = "SO100HandOverBanana", cameras: tuple = (), camera_resolution: tuple = (480, 640),
                 time_limit: float = 30.0, device: str = "cuda:0", n_envs: int = 1, **env_kwargs):
        from . import task_suite
        if cameras:
            raise NotImplementedError("camera observations are outside the MI355X hot path (SURVEY.md 8b); pass cameras=()")
        self.device = device
        self.cameras = tuple(cameras)
        self.camera_resolution = camera_resolution
        self.n_envs = int(n_envs)
        self.env = task_suite.create_task_env(task_name=task_name, time_limit=time_limit, cameras=(), n_envs=self.n_envs,
                                              device=device, **env_kwargs)
        self.episode_index = 0
        self.frame_index = 0
        self.start_time = 0.0

    # -- reference surface ---------------------------------------------------------------------------------
    def reset(self) -> Dict[str, Any]:
        self.frame_index = 0
        self.start_time = 0.0
        if self.n_envs == 1:
            ts = self.env.reset()
            return to_lerobot_format(ts.observation["joints_pos"], None, 0, self.episode_index, self.device)
        self.env.reset_all()
        return to_lerobot_format(self.env.obs[:, 0:6].clone(), None, 0, self.episode_index, self.device, self.n_envs)

    def step(self, action) -> Dict[str, Any]:
        if self.n_envs == 1:
            ts = self.env.step(action)
            self.frame_index += 1
            return to_lerobot_format(ts.observation["joints_pos"], np.asarray(action), self.frame_index, self.episode_index,
                                     self.device)
        import torch
        act = torch.as_tensor(action, dtype=torch.float32, device=self.env.device)
        obs, _, _, _ = self.env.step_tensor(act)
        self.frame_index += 1
        return to_lerobot_format(obs[:, 0:6].clone(), act.clone(), self.frame_index, self.episode_index, self.device, self.n_envs)

    def collect_episode(self, actions: List[Any], save_path: Optional[str] = None) -> List[Dict[str, Any]]:
        episode = [self.reset()]
        for a in actions:
            episode.append(self.step(a))
        if save_path:
            import torch
            torch.save(episode, save_path)
        self.episode_index += 1
        return episode

    def get_action_spec(self) -> Dict[str, Any]:
        return {"shape": (6,), "dtype": np.float32, "low": -1.0, "high": 1.0, "names": list(JOINT_ACTION_NAMES)}

    def get_observation_spec(self) -> Dict[str, Any]:
        return {"images": {}, "state": {"shape": (6,), "dtype": np.float32, "names": [f"joint_{i}" for i in range(6)]}}
