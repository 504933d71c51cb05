"""ALOHA hand-over tasks (`HandOverBanana`, `HandOverPen`) on the general-tree engine.

Host-side mirror of the reference's `HandOver` task (so101_sim/tasks/hand_over.py:122-349) on `AlohaTask`
(so101_sim/tasks/base/aloha2_task.py:145-444) behind the same dm_env-style surface as the SO100 environments of env.py:
    env.reset() -> TimeStep    env.step(action[14]) -> TimeStep    env.action_spec()    env.observation_spec()
    env.task.get_instruction()    env.close()
Every number comes from the kernels behind the `so101_tree_*` entry points of include/so101.h (csrc/so101_tree.hpp); torch is the
array container.  n_envs == 1 yields numpy observations without the env dimension (what a caller of the reference sees), n_envs > 1
torch tensors on the GPU with a leading env dimension.

Not built: cameras (no renderer in this library) and a non-default table height offset (the committed model blob is compiled for
the reference's default).  The observation delays ARE parameters (`joints_observation_delay_secs`, `image_observation_delay_secs`,
aloha2_task.py:153-159: whole control steps, handed to the kernels by so101_tree_configure_env), and `physics_state` /
`delayed_physics_state` come from a device-side delay line (so101_tree_bind_physics_state).  Both reward modes are built: the overlap boxes (default) and the contact sequence
(`reward_based_on_overlap=False`, hand_over.py:286-338, with `reward_requires_handover`).
"""
from __future__ import annotations

import collections
import os

import numpy as np

from . import native
from ._dmenv import Array, BoundedArray, TimeStep
from .model import blob as blobfmt
from .model import scenes

DEFAULT_CONTROL_TIMESTEP = 0.02
PHYSICS_TIMESTEP = 0.002
DEFAULT_JOINTS_DELAY_SECS, DEFAULT_PHYSICS_DELAY_SECS = 0.1, 0.3          # aloha2_task.py:102-103
NPOS, NVEL = 14, 16

# obs row of so101_tree_step (include/so101.h): joints_pos | joints_vel | undelayed_joints_pos | undelayed_joints_vel | commanded_joints_pos
_SLICES = dict(joints_pos=(0, 14), joints_vel=(14, 30), undelayed_joints_pos=(30, 44), undelayed_joints_vel=(44, 60),
               commanded_joints_pos=(60, 74))


class HandOverTask:
    """Host-side description of `HandOver` (hand_over.py:128-236)."""

    def __init__(self, object_name, reward_based_on_overlap=True, reward_requires_handover=False, **kwargs):
        if object_name not in scenes.HANDOVER_CONFIGS:
            raise ValueError(f"Invalid object name: {object_name}, must be one of {scenes.HANDOVER_CONFIGS.keys()}")
        self.object_name = object_name
        self.reward_based_on_overlap = bool(reward_based_on_overlap)         # False: the contact sequence of hand_over.py:286-338
        self.reward_requires_handover = bool(reward_requires_handover)       # (read by the contact-sequence mode only, as in the reference)
        self.control_timestep = float(kwargs.pop("control_timestep", DEFAULT_CONTROL_TIMESTEP))
        self.cameras = tuple(kwargs.pop("cameras", ()))
        self.image_observation_enabled = bool(kwargs.pop("image_observation_enabled", True))
        self.terminate_episode = bool(kwargs.pop("terminate_episode", True))
        self.waist_joint_limit = float(kwargs.pop("waist_joint_limit", np.pi / 2))
        if float(kwargs.pop("table_height_offset", scenes.ALOHA_TABLE_HEIGHT_OFFSET)) != scenes.ALOHA_TABLE_HEIGHT_OFFSET:
            raise NotImplementedError("the model blob is compiled for table_height_offset = 0.011 (aloha2_task.py:107)")
        # observation delays (aloha2_task.py:153-159,236-251): dm_control delays an observable by delay_secs / physics_timestep
        # substeps; the kernels keep one sample per control step, so a delay must be a whole number of control steps (0 = undelayed)
        self.joints_observation_delay_secs = float(kwargs.pop("joints_observation_delay_secs", DEFAULT_JOINTS_DELAY_SECS))
        self.image_observation_delay_secs = float(kwargs.pop("image_observation_delay_secs", DEFAULT_PHYSICS_DELAY_SECS))
        self.joints_delay_steps = self._delay_steps(self.joints_observation_delay_secs, "joints_observation_delay_secs")
        self.physics_delay_steps = self._delay_steps(self.image_observation_delay_secs, "image_observation_delay_secs")
        self._instruction = scenes.ALOHA_INSTRUCTIONS[object_name]

    scene = "hand_over"

    def load_blob(self, real: str):
        return scenes.load_aloha_blob(self.object_name, real)

    @property
    def reward_mode(self) -> int:          # so101_tree_config.reward_mode
        return 0 if self.reward_based_on_overlap else 1

    def _delay_steps(self, secs: float, name: str) -> int:
        steps = secs / self.control_timestep
        if secs < 0 or abs(steps - round(steps)) > 1e-9 or round(steps) > 64:
            raise ValueError(f"{name} = {secs}: must be a whole number (0..64) of control steps of {self.control_timestep} s")
        return int(round(steps))

    def get_instruction(self):
        return self._instruction


class DiningPlaceInContainerTask(HandOverTask):
    """Host-side description of `DiningPlaceInContainer` (tasks/dining_place_in_container.py:26-160) on the `Dining` scene
    (tasks/base/dining.py:39-267): the ALOHA robot and six free props (mug, pen, banana, plate, bowl, container); `task_id` names the
    object / receptacle pair and its reward - 'banana' (into the bowl) and 'pen' (into the white cup): overlap boxes; 'mug' (onto the
    plate): the mug touches the plate while neither moves."""

    scene = "dining"

    def __init__(self, task_id: str = "banana", **kwargs):
        if task_id not in scenes.DINING_TASKS:
            raise ValueError(f"Unknown task ID: {task_id}")                    # dining_place_in_container.py:83-84
        cfg = scenes.DINING_TASKS[task_id]
        self.task_id = task_id
        self.object_name = cfg["object"]
        self.receptacle_name = cfg["receptacle"]
        self.reward_type = cfg["reward"]
        self.reward_based_on_overlap = cfg["reward"] == "bbox"
        self.reward_requires_handover = False
        self.control_timestep = float(kwargs.pop("control_timestep", DEFAULT_CONTROL_TIMESTEP))
        self.cameras = tuple(kwargs.pop("cameras", ()))
        self.image_observation_enabled = bool(kwargs.pop("image_observation_enabled", True))
        self.terminate_episode = bool(kwargs.pop("terminate_episode", True))
        self.waist_joint_limit = float(kwargs.pop("waist_joint_limit", np.pi / 2))
        if float(kwargs.pop("table_height_offset", scenes.ALOHA_TABLE_HEIGHT_OFFSET)) != scenes.ALOHA_TABLE_HEIGHT_OFFSET:
            raise NotImplementedError("the model blob is compiled for table_height_offset = 0.011 (aloha2_task.py:107)")
        self.joints_observation_delay_secs = float(kwargs.pop("joints_observation_delay_secs", DEFAULT_JOINTS_DELAY_SECS))
        self.image_observation_delay_secs = float(kwargs.pop("image_observation_delay_secs", DEFAULT_PHYSICS_DELAY_SECS))
        self.joints_delay_steps = self._delay_steps(self.joints_observation_delay_secs, "joints_observation_delay_secs")
        self.physics_delay_steps = self._delay_steps(self.image_observation_delay_secs, "image_observation_delay_secs")
        self._instruction = cfg["instruction"]

    def load_blob(self, real: str):
        return scenes.load_dining_blob(self.task_id, real)

    @property
    def reward_mode(self) -> int:
        return scenes.DINING_REWARD_MODE[self.reward_type]


def aloha_action_spec(ctrlrange: np.ndarray, waist_joint_limit: float = np.pi / 2) -> BoundedArray:
    """AlohaTask.action_spec (aloha2_task.py:279-301): the actuators' ctrlrange, waists cut to +-waist_joint_limit, grippers in
    follower units; shape (14,), float32."""
    lo, hi = ctrlrange[:, 0].astype(np.float32), ctrlrange[:, 1].astype(np.float32)
    lo[0] = lo[7] = -waist_joint_limit
    hi[0] = hi[7] = waist_joint_limit
    lo[6] = lo[13] = scenes.ALOHA_GRIPPER_LIMITS["follower"][1]
    hi[6] = hi[13] = scenes.ALOHA_GRIPPER_LIMITS["follower"][0]
    return BoundedArray((14,), np.float32, lo, hi)


class AlohaEnvironment:
    def __init__(self, task: HandOverTask, n_envs: int = 1, time_limit: float = float("inf"), random_state=None, device=None,
                 env_id_base: int = 0, solver_iterations: int = 0, solver_tolerance: float = -1.0, settle_max_substeps: int = 1000,
                 physics_state: bool | None = None, seed_compatible: bool = True, narrowphase: str = "epa", prefetch_resets: bool = True, pipeline: bool = True):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("so101_sim_amd needs a ROCm GPU (MI355X): the step path has no CPU fallback")
        self.torch = torch
        self.task = task
        self.n_envs = N = int(n_envs)
        self.device = torch.device(device if device is not None else "cuda:0")
        # N = 1: the reference's placements come from np.random.RandomState(seed) inside dm_control's PropPlacers.  With
        # seed_compatible (default) a single env draws them from the same generator in the same order - object position (3 draws,
        # hand_over.py:36-40), object yaw (1, :41-49), container position (3 per attempt, <= 20 attempts against collisions, :51-56) -
        # and lets the kernels settle (so101_tree_settle).  Batches key the kernels' counter RNG by (seed, env id, episode) instead.
        self._seed_compatible = bool(seed_compatible) and int(n_envs) == 1
        self._np_random = random_state if isinstance(random_state, np.random.RandomState) else (
            np.random.RandomState() if random_state is None else np.random.RandomState(int(random_state)))
        self._pending_first = False
        if isinstance(random_state, np.random.RandomState):
            seed = 0 if self._seed_compatible else int(random_state.randint(0, 2**31 - 1))
        elif random_state is None:
            seed = int.from_bytes(os.urandom(4), "little")
        else:
            seed = int(random_state)
        self.seed = seed
        blob, self.meta = task.load_blob("f32")
        m = blobfmt.unpack(blob)
        self._ctrlrange = np.asarray(m["act_ctrlrange"], dtype=np.float64).reshape(-1, 2)
        con_body, obj_body = int(np.asarray(m["task_container_body"]).ravel()[0]), int(np.asarray(m["task_object_body"]).ravel()[0])
        qadr = np.asarray(m["body_qposadr"])
        m64 = blobfmt.unpack(task.load_blob("f64")[0])      # (the distributions' bounds unrounded: the draws must be the reference's, bit for bit)
        if task.scene == "dining":
            self._dining = dict(lo=np.asarray(m64["task_region_lo"], dtype=np.float64).reshape(6, 3), hi=np.asarray(m64["task_region_hi"], dtype=np.float64).reshape(6, 3),
                                qadr=[int(qadr[b]) for b in np.asarray(m["task_prop_bodies"]).ravel()])
        self._placer = dict(obj_lo=np.asarray(m64["task_obj_pos_lo"], dtype=np.float64), obj_hi=np.asarray(m64["task_obj_pos_hi"], dtype=np.float64),
                            yaw=np.asarray(m64["task_obj_yaw"], dtype=np.float64), con_lo=np.asarray(m64["task_con_pos_lo"], dtype=np.float64),
                            con_hi=np.asarray(m64["task_con_pos_hi"], dtype=np.float64), qo=int(qadr[obj_body]), qc=int(qadr[con_body]),
                            con_geoms=set(np.nonzero(np.asarray(m["geom_body"]) == con_body)[0].tolist()))
        with torch.cuda.device(self.device):
            if narrowphase not in ("mpr", "epa"):
                raise ValueError(f"narrowphase must be 'mpr' or 'epa', got {narrowphase!r}")
            from . import build as _build
            self.narrowphase = narrowphase       # "mpr": the -DSO101_MPR build of the library, built on demand (env.py)
            self.sim = native.TreeSim(blob, N, device=self.device.index or 0, lib_path=_build.build(mpr=True) if narrowphase == "mpr" else None)
        s = self.sim
        if (s.nu, s.obs_dim) != (NPOS, 3 * NPOS + 2 * NVEL):
            raise RuntimeError("unexpected model dimensions for an ALOHA scene")
        z = lambda *sh, dt=torch.float32: torch.zeros(*sh, dtype=dt, device=self.device)
        self.qpos, self.qvel, self.ctrl, self.warm = z(s.nq, N), z(s.nv, N), z(s.nu, N), z(s.nv, N)
        jd, pd = task.joints_delay_steps, task.physics_delay_steps
        self._ring_pos, self._ring_vel = z(max(jd, 1), NPOS, N), z(max(jd, 1), NVEL, N)
        self.ep_return, self.step_count, self.episode = z(N), z(N, dt=torch.int32), z(N, dt=torch.int32)
        self.obs, self.reward, self.discount = z(N, s.obs_dim), z(N), z(N)
        self.step_type = z(N, dt=torch.uint8)
        self._action = z(N, s.nu)
        self._diag = z(N, 8, dt=torch.int32)
        s.bind(*(t.data_ptr() for t in (self.qpos, self.qvel, self.ctrl, self.warm)))
        s.bind_env(*(t.data_ptr() for t in (self._ring_pos, self._ring_vel, self.ep_return, self.step_count, self.episode)))
        nsub = int(round(task.control_timestep / PHYSICS_TIMESTEP))
        self.last_step = scenes.time_limit_last_step(time_limit, task.control_timestep, PHYSICS_TIMESTEP) if np.isfinite(time_limit) else 1 << 30
        s.configure_env(n_substeps=nsub, last_step=self.last_step, settle_max_substeps=int(settle_max_substeps),
                        terminate_on_success=int(task.terminate_episode), solver_iterations=int(solver_iterations),
                        solver_tolerance=float(solver_tolerance), seed=seed, env_id_base=int(env_id_base),
                        reward_mode=task.reward_mode, reward_requires_handover=int(task.reward_requires_handover),
                        joints_delay_steps=jd, physics_delay_steps=pd,
                        # reset prefetch (batches; a seed-compatible single env draws its placements on the host: nothing to settle ahead)
                        prefetch_resets=int(bool(prefetch_resets) and not self._seed_compatible),
                        # the step as a launch chain (narrowphase in a launch of its own; the contact rewards end the chain with one more narrowphase
                        # launch on the post-step state + k_tree_pipe_finish); one env is one wavefront either way and keeps the single launch
                        pipeline=int(bool(pipeline) and self.n_envs > 1))
        self._pipeline = bool(pipeline)
        # physics_state / delayed_physics_state (aloha2_task.py:244-251,441-444): qpos | qvel and its copy of `pd` control steps ago,
        # from a device-side delay line the step / reset kernels maintain.  The reference ties them to image_observation_enabled;
        # for batches they are opt-in (58 + 58 floats per env and step).
        self._with_state = bool(task.image_observation_enabled if physics_state is None and N == 1 else physics_state)
        self._ps_dim = s.nq + s.nv
        self.physics_state = self.delayed_physics_state = self._ps_ring = None
        if self._with_state:
            self._ps_ring = z(max(pd, 1), self._ps_dim, N)
            self.physics_state, self.delayed_physics_state = z(N, self._ps_dim), z(N, self._ps_dim)
            s.bind_physics_state(self._ps_ring.data_ptr(), self.physics_state.data_ptr(), self.delayed_physics_state.data_ptr())

    # ------------------------------------------------------------------ specs
    def action_spec(self) -> BoundedArray:
        return aloha_action_spec(self._ctrlrange, self.task.waist_joint_limit)

    def observation_spec(self):
        spec = collections.OrderedDict()
        spec["commanded_joints_pos"] = Array((NPOS,), np.float64, "commanded_joints_pos")
        spec["joints_pos"] = Array((NPOS,), np.float64, "joints_pos")
        spec["joints_vel"] = Array((NVEL,), np.float64, "joints_vel")
        if self._with_state:
            spec["physics_state"] = Array((self._ps_dim,), np.float64, "physics_state")
        if self.task.joints_delay_steps:          # (the undelayed_* / delayed_* copies exist only with a joints delay, aloha2_task.py:236-243)
            spec["undelayed_joints_pos"] = Array((NPOS,), np.float64, "undelayed_joints_pos")
            spec["undelayed_joints_vel"] = Array((NVEL,), np.float64, "undelayed_joints_vel")
            spec["delayed_joints_pos"] = Array((NPOS,), np.float64, "delayed_joints_pos")
            spec["delayed_joints_vel"] = Array((NVEL,), np.float64, "delayed_joints_vel")
        if self._with_state and self.task.physics_delay_steps:
            spec["delayed_physics_state"] = Array((self._ps_dim,), np.float64, "delayed_physics_state")
        return spec

    # ------------------------------------------------------------------ stepping
    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def _obs_dict(self):
        o = collections.OrderedDict()
        cut = lambda k: self.obs[:, _SLICES[k][0]:_SLICES[k][1]]
        o["commanded_joints_pos"] = cut("commanded_joints_pos")
        o["joints_pos"] = cut("joints_pos")
        o["joints_vel"] = cut("joints_vel")
        if self._with_state:
            o["physics_state"] = self.physics_state
        if self.task.joints_delay_steps:
            o["undelayed_joints_pos"] = cut("undelayed_joints_pos")
            o["undelayed_joints_vel"] = cut("undelayed_joints_vel")
            o["delayed_joints_pos"] = cut("joints_pos")
            o["delayed_joints_vel"] = cut("joints_vel")
        if self._with_state and self.task.physics_delay_steps:
            o["delayed_physics_state"] = self.delayed_physics_state
        if self.n_envs == 1:
            return collections.OrderedDict((k, v[0].double().cpu().numpy()) for k, v in o.items())
        return o

    def step_tensor(self, action):
        """action: float tensor [N, 14] on the device.  Fills and returns (obs, reward, discount, step_type)."""
        if tuple(action.shape) != (self.n_envs, NPOS):
            raise ValueError(f"Expected 14 joint positions per env, got {tuple(action.shape)}")
        self._action.copy_(action)
        self.sim.step(self._action.data_ptr(), self.obs.data_ptr(), self.reward.data_ptr(), self.discount.data_ptr(),
                      self.step_type.data_ptr(), self._stream())
        return self.obs, self.reward, self.discount, self.step_type

    def _timestep(self):
        obs = self._obs_dict()
        if self.n_envs == 1:
            st = int(self.step_type[0])
            if st == 0:
                return TimeStep(st, None, None, obs)
            return TimeStep(st, float(self.reward[0]), float(self.discount[0]), obs)
        return TimeStep(self.step_type, self.reward, self.discount, obs)

    def _container_collides(self) -> bool:
        if not hasattr(self, "_dbg"):
            self._dbg = self.torch.zeros(1, self.sim.debug_dim, device=self.device)
        self.sim.debug_forward(self._dbg.data_ptr(), self._stream())
        r = self._dbg[0].cpu().numpy()
        D = self.sim.dbg
        for k in range(int(r[D["COUNTS"]])):
            if int(r[D["CON"] + 10 * k + 7]) in self._placer["con_geoms"] or int(r[D["CON"] + 10 * k + 8]) in self._placer["con_geoms"]:
                return True
        return False

    def _reset_seed_compatible_dining(self):
        """Dining._sample_props + the two PropPlacers (dining.py:162-267) with numpy's generator, in the reference's order: six region
        samples `uniform(low, high)` (top left / middle / right, bottom left / middle / right: three numbers each), `shuffle` of the top and
        of the bottom ordering, then per prop in the order plate, bowl, container, mug, pen, banana one `uniform(-pi, pi)` yaw (positions
        come from a deterministic Sequence); collisions ignored; settle and episode start by the kernels."""
        torch, D, rs, s = self.torch, self._dining, self._np_random, self.sim
        samples = [rs.uniform(low=D["lo"][r], high=D["hi"][r]) for r in range(6)]
        top, bottom = [0, 1, 2], [3, 4, 5]
        rs.shuffle(top)
        rs.shuffle(bottom)
        regions = [top[0], top[1], top[2], bottom[0], bottom[1], bottom[2]]          # plate, bowl, container, mug, pen, banana
        q = np.zeros(s.nq)
        q[:16] = np.concatenate([scenes.ALOHA_HOME_QPOS, scenes.ALOHA_HOME_QPOS])
        yaws = []
        for p in range(6):
            yaw = rs.uniform(-np.pi, np.pi)
            a = D["qadr"][p]
            q[a:a + 3] = samples[regions[p]]
            q[a + 3:a + 7] = [np.cos(0.5 * yaw), 0.0, 0.0, np.sin(0.5 * yaw)]
            yaws.append(float(yaw))
        self.ctrl.copy_(torch.as_tensor(np.concatenate([scenes.ALOHA_HOME_CTRL, scenes.ALOHA_HOME_CTRL]), dtype=torch.float32, device=self.device).unsqueeze(1))
        self.qpos.copy_(torch.as_tensor(q, dtype=torch.float32, device=self.device).unsqueeze(1))
        self.qvel.zero_(); self.warm.zero_()
        self.placements = dict(zip(scenes.DINING_PLACER_ORDER, [dict(position=samples[regions[p]].copy(), yaw=yaws[p], region=int(regions[p])) for p in range(6)]))
        s.settle(self._stream())
        s.begin_episode(self._stream())
        self.episode += 1
        if int(self.diagnostics()[0, 4]) & 32:
            import warnings
            warnings.warn("Failed to settle physics within the settle budget (dm_control warns likewise)")

    def _reset_seed_compatible(self):
        """placements from numpy's generator in dm_control's PropPlacer order, settle and episode start by the kernels"""
        if self.task.scene == "dining":
            return self._reset_seed_compatible_dining()
        torch, P, rs, s = self.torch, self._placer, self._np_random, self.sim
        opos = rs.uniform(P["obj_lo"], P["obj_hi"])
        yaw = rs.uniform(P["yaw"][0], P["yaw"][1])
        q = np.zeros(s.nq)
        q[:16] = np.concatenate([scenes.ALOHA_HOME_QPOS, scenes.ALOHA_HOME_QPOS])           # aloha2_task.py:374-377
        q[P["qo"]:P["qo"] + 3] = opos
        q[P["qo"] + 3:P["qo"] + 7] = [np.cos(0.5 * yaw), 0.0, 0.0, np.sin(0.5 * yaw)]
        q[P["qc"] + 3] = 1.0
        self.ctrl.copy_(torch.as_tensor(np.concatenate([scenes.ALOHA_HOME_CTRL, scenes.ALOHA_HOME_CTRL]), dtype=torch.float32, device=self.device).unsqueeze(1))
        placed = False
        for _ in range(20):                                                                  # PropPlacer max_attempts_per_prop
            q[P["qc"]:P["qc"] + 3] = rs.uniform(P["con_lo"], P["con_hi"])
            self.qpos.copy_(torch.as_tensor(q, dtype=torch.float32, device=self.device).unsqueeze(1))
            self.qvel.zero_(); self.warm.zero_()
            if not self._container_collides():
                placed = True
                break
        if not placed:
            raise RuntimeError("Failed to place the container without collisions in 20 attempts (dm_control PropPlacer raises here too)")
        self.placements = dict(object_position=opos.copy(), object_yaw=float(yaw), container_position=q[P["qc"]:P["qc"] + 3].copy())
        s.settle(self._stream())
        s.begin_episode(self._stream())
        self.episode += 1
        if int(self.diagnostics()[0, 4]) & 32:
            import warnings
            warnings.warn("Failed to settle physics within the settle budget (dm_control warns likewise)")

    def reset(self) -> TimeStep:
        """every env starts a new episode; returns FIRST"""
        if self._seed_compatible:
            self._reset_seed_compatible()
        else:
            self.sim.reset(None, self._stream())
        self._pending_first = False
        self.torch.cuda.current_stream(self.device).synchronize()
        # the FIRST observation is written by the step kernel for an env that resets inside a step call; for an explicit reset it
        # is assembled here from the state the reset left
        q = self.qpos.t()
        pos = q[:, [0, 1, 2, 3, 4, 5, 6, 8, 9, 10, 11, 12, 13, 14]].clone()
        (qo, qc), (_, _), (fo, fc) = scenes.ALOHA_GRIPPER_LIMITS["sim_qpos"], scenes.ALOHA_GRIPPER_LIMITS["sim_ctrl"], scenes.ALOHA_GRIPPER_LIMITS["follower"]
        for k in (6, 13):
            pos[:, k] = (pos[:, k] - qc) / (qo - qc) * (fo - fc) + fc
        cmd = self.ctrl.t().clone()
        (co, cc) = scenes.ALOHA_GRIPPER_LIMITS["sim_ctrl"]
        for k in (6, 13):
            cmd[:, k] = (cmd[:, k] - cc) / (co - cc) * (fo - fc) + fc
        vel = self.qvel.t()[:, :NVEL]
        self.obs[:, 0:14], self.obs[:, 14:30], self.obs[:, 30:44], self.obs[:, 44:60], self.obs[:, 60:74] = pos, vel, pos, vel, cmd
        self.step_type.zero_(); self.reward.zero_(); self.discount.fill_(1.0)
        ts = self._timestep()
        return TimeStep(ts.step_type, None, None, ts.observation)       # FIRST carries no reward / discount

    def step(self, action) -> TimeStep:
        torch = self.torch
        a = torch.as_tensor(np.asarray(action) if not torch.is_tensor(action) else action, dtype=torch.float32, device=self.device)
        if a.dim() == 1:
            a = a.unsqueeze(0)
        if self._pending_first:                # the step after LAST restarts the episode and reports FIRST (dm_control); the host draws
            return self.reset()                # the next placements, the kernels' own auto-reset is not used
        self.step_tensor(a)
        if self._seed_compatible and int(self.step_type[0]) == 2:
            self._pending_first = True
        return self._timestep()

    def compute_settled(self, n_episodes: int, first_episode: int = 0):
        """Placement + settle of the first episodes of every env, done once at the width of the machine (one launch per episode) and
        attached: the resets of those episodes - explicit or inside step calls - become copies, bit-identical to settling in place.
        Batches only (a seed-compatible single env draws its placements on the host)."""
        if self._seed_compatible:
            raise RuntimeError("a seed-compatible single env draws its placements from numpy's generator; nothing to precompute")
        torch, s, N = self.torch, self.sim, self.n_envs
        z = lambda *sh, dt=torch.float32: torch.zeros(*sh, dtype=dt, device=self.device)
        self._store = (z(n_episodes, s.nq, N), z(n_episodes, s.nv, N), z(n_episodes, s.nv, N), z(n_episodes, N, dt=torch.int32))
        s.compute_settled(first_episode, n_episodes, *(t.data_ptr() for t in self._store), self._stream())
        self.torch.cuda.current_stream(self.device).synchronize()
        s.set_settled_store(first_episode, n_episodes, *(t.data_ptr() for t in self._store))
        return self._store

    def episode_returns(self):
        return self.ep_return

    def launch_plan(self) -> dict:
        """What the last `step_tensor` call enqueued, as the LIBRARY counted it (`so101_tree_last_plan`; round 5 re-implemented the slicing rule
        of csrc/tu_tree.hip here, which a change over there or a late SO101_TREE_SLICES would silently falsify - ADVICE r5).  Launch chain:
        per env slice one memset, k_tree_pipe_begin, n_substeps x (k_tree_narrow + k_tree_pipe_solve), and for the contact rewards one more
        k_tree_narrow + k_tree_pipe_finish."""
        slices, launches, memsets, path = self.sim.last_plan()
        if path == 0:
            return {"path": "no step yet", "slices": 0, "kernel_launches": 0, "memsets": 0}
        if path == 1:
            return {"path": "single kernel (k_tree_step)", "slices": 1, "kernel_launches": launches, "memsets": memsets}
        nsub = int(round(self.task.control_timestep / PHYSICS_TIMESTEP))
        post = self.task.reward_mode != 0
        return {"path": "launch chain", "slices": slices, "kernel_launches": launches, "memsets": memsets,
                "kernels": "k_tree_pipe_begin + %d x (k_tree_narrow + k_tree_pipe_solve)%s per slice" % (nsub, " + k_tree_narrow + k_tree_pipe_finish" if post else "")}

    def diagnostics(self):
        """per env: contacts, constraint rows, solver iterations, broadphase candidates, flags of the last substep"""
        self.sim.get_diag(self._diag.data_ptr(), self._stream())
        return self._diag

    def close(self):
        self.sim.close()
