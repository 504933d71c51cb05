"""Batched dm_env-style environment over the HIP step library.

Mirrors the surface callers of the reference use:
    env.reset() -> TimeStep         env.step(action) -> TimeStep        env.action_spec()
    env.observation_spec()          env.close()         env.task.get_instruction()     env.physics
`composer.Environment` semantics kept: FIRST has reward/discount None; a step after LAST resets and
returns FIRST; no bounds validation of actions; wrong action length raises ValueError
(scripts/so101_calibration.py:71-72).

N == 1 (`SingleEnvironment`, what `create_task_env` returns by default) yields numpy observations
with exactly the reference's keys and shapes; N > 1 (`BatchedEnvironment`) yields torch tensors on
the GPU with a leading env dimension.  PyTorch is only the array container: every number is produced
by the kernels behind include/so101.h.
"""
from __future__ import annotations

import collections
import hashlib
import os

import numpy as np

from . import native
from ._dmenv import BoundedArray, Array, StepType, TimeStep
from .calibration import SO101Calibration
from .model import scenes

DEFAULT_CONTROL_TIMESTEP = 0.02
PHYSICS_TIMESTEP = 0.002
_JOINT_DELAY_STEPS = 5       # 0.1 s / 0.02 s  (so100_task.py:81,196-201)
_PHYSICS_DELAY_STEPS = 15    # 0.3 s / 0.02 s  (so100_task.py:80,204-210)


def so100_action_spec(rotation_joint_limit: float = np.pi) -> BoundedArray:
    """SO100Task.action_spec (so100_task.py:232-251): ctrlrange of scene_pbr.xml:11 (+-3.14158) with
    [0] := +-rotation_joint_limit and [5] := [0, 0.08]; shape (6,), float32."""
    lo = np.full(6, -3.14158, dtype=np.float32)
    hi = np.full(6, 3.14158, dtype=np.float32)
    lo[0], hi[0] = -rotation_joint_limit, rotation_joint_limit
    lo[5], hi[5] = 0.0, 0.08
    return BoundedArray((6,), np.float32, lo, hi)


OBSERVATION_KEYS = ("commanded_joints_pos", "joints_pos", "joints_vel", "physics_state", "undelayed_joints_pos",
                    "undelayed_joints_vel", "delayed_physics_state")   # examples/so101_rl_breakdown.ipynb:65 (cameras excluded)


class SO100HandOverTask:
    """Host-side description of `SO100HandOver` (so101_sim/tasks/so100_hand_over.py:121-236)."""

    def __init__(self, object_name, reward_based_on_overlap=True, **kwargs):
        if object_name not in scenes.HANDOVER_CONFIGS:
            raise ValueError(f"Invalid object name: {object_name}, must be one of {scenes.HANDOVER_CONFIGS.keys()}")
        if not reward_based_on_overlap:
            raise NotImplementedError(
                "contact+distance reward mode (so100_hand_over.py:277-318) is not built: in the reference that branch looks up "
                "a body 'so100/hand_link' which scene_pbr.xml does not define, so it raises there as well")
        self.object_name = object_name
        self.control_timestep = float(kwargs.pop("control_timestep", DEFAULT_CONTROL_TIMESTEP))
        self.cameras = tuple(kwargs.pop("cameras", ()))
        self.image_observation_enabled = bool(kwargs.pop("image_observation_enabled", True))
        self.terminate_episode = bool(kwargs.pop("terminate_episode", True))
        self.rotation_joint_limit = float(kwargs.pop("rotation_joint_limit", np.pi))
        self._instruction = scenes.HANDOVER_CONFIGS[object_name]["instruction"]
        # the calibration file is looked up relative to the CWD at construction time, as in the reference
        self.calibration = SO101Calibration()

    def get_instruction(self):
        return self._instruction


class _PhysicsView:
    """Partial stand-in for `env.physics`: exposes `.data.qpos/.qvel/.ctrl` views of env 0 and
    `.time()`; rendering is out of scope."""

    class _Data:
        pass

    def __init__(self, env):
        self._env = env
        self.data = self._Data()

    def _refresh(self):
        e = self._env
        self.data.qpos = e.qpos[:, 0].detach().cpu().numpy().astype(np.float64)
        self.data.qvel = e.qvel[:, 0].detach().cpu().numpy().astype(np.float64)
        self.data.ctrl = e.ctrl[:, 0].detach().cpu().numpy().astype(np.float64)
        return self

    def time(self):
        return float(self._env.step_count[0].item()) * self._env.task.control_timestep

    def get_state(self):
        self._refresh()
        return np.concatenate([self.data.qpos, self.data.qvel])

    def render(self, *a, **k):
        raise NotImplementedError("rendering is outside the MI355X hot path (SURVEY.md 8b)")


class BatchedEnvironment:
    def __init__(self, task: SO100HandOverTask, n_envs: int = 1, time_limit: float = float("inf"),
                 random_state=None, device=None, env_id_base: int = 0, solver_iterations: int = 0,
                 solver_tolerance: float = -1.0, settle_max_substeps: int = 1000, solver: str = "newton",
                 prefetch_resets: bool = True, physics_state: bool = False, narrowphase: str = "epa"):
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("so101_sim_amd needs a ROCm GPU (MI355X): the step path has no CPU fallback")
        self.torch = torch
        self.task = task
        self.n_envs = int(n_envs)
        self.device = torch.device(device if device is not None else "cuda:0")
        if isinstance(random_state, np.random.RandomState):
            seed = int(random_state.randint(0, 2**31 - 1))
        elif random_state is None:
            seed = int.from_bytes(os.urandom(4), "little")
        else:
            seed = int(random_state)
        self.seed = seed
        blob, self.meta = scenes.load_blob(task.object_name, "f32")
        dev_index = self.device.index or 0
        with torch.cuda.device(self.device):
            # narrowphase = "epa" (default): the penetration of non-flat convex pairs is the minimum translation (MPR's final portal
            # expanded by EPA to the nearest face of the Minkowski difference, as mujoco >= 3.3's native GJK / EPA reports it - the
            # reference pins mujoco>=3.3.3, requirements.txt:7).  "mpr": the -DSO101_MPR build of the library (built on demand) keeps
            # MPR's portal depth: 5-15 % faster, 3x the physics errors under random actions (profiles/README.md)
            if narrowphase not in ("mpr", "epa"):
                raise ValueError(f"narrowphase must be 'mpr' or 'epa', got {narrowphase!r}")
            from . import build as _build
            self.narrowphase = narrowphase
            # (SO101_HIP_LIB - kernel experiments, bench.py --pipeline 2|3 - names the library whatever the narrowphase: whoever sets it
            #  built the variant it wants, e.g. build(exp=True, mpr=True); an explicit lib_path here used to override it)
            import os as _os
            lib = None if (_os.environ.get("SO101_HIP_LIB") or narrowphase != "mpr") else _build.build(mpr=True)
            self.sim = native.Sim(blob, self.n_envs, device=dev_index, seed=seed, lib_path=lib)
        N = self.n_envs
        z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=self.device)
        self.qpos, self.qvel, self.ctrl, self.warm = z(20, N), z(18, N), z(6, N), z(18, N)
        self.ring, self.ep_return = z(native.RING_DEPTH, 6, N), z(N)
        self.step_count, self.episode = z(N, dt=torch.int32), z(N, dt=torch.int32)
        self.obs, self.reward, self.discount = z(N, native.OBS_DIM), z(N), z(N)
        self.step_type = z(N, dt=torch.uint8)
        self._pack = None
        if N == 1:
            # ONE env (the reference's own use: run_eval.py, the notebooks): state and per-step outputs are views of one 256-byte buffer, so that a
            # step hands everything the numpy observation needs back to the host in ONE copy into pinned memory (round 4: six device-to-host
            # round trips per step, 0.45 ms of the 2.2 ms)
            self._pack = z(64)
            self.qpos, self.qvel = self._pack[0:20].view(20, 1), self._pack[20:38].view(18, 1)
            self.obs, self.reward, self.discount = self._pack[38:56].view(1, 18), self._pack[56:57], self._pack[57:58]
            self.step_type = self._pack[58:59].view(torch.uint8)[0:1]
            self._host = torch.zeros(64, dtype=torch.float32).pin_memory()
            self._act_host = torch.zeros(1, native.ACT_DIM, dtype=torch.float32).pin_memory()
        self._action = z(N, native.ACT_DIM)
        # per-env mass / inertia scale of (object, container): domain randomisation, 1.0 = the model's props
        self.mass_scale = torch.ones(2, N, dtype=torch.float32, device=self.device)
        self._events = torch.zeros(native.NEVENTS, dtype=torch.int64, device=self.device)
        self._pool = None
        self.sim.bind(*(t.data_ptr() for t in (self.qpos, self.qvel, self.ctrl, self.warm, self.ring,
                                                 self.ep_return, self.step_count, self.episode, self.mass_scale)))
        # physics_state / delayed_physics_state (so100_task.py:203-210,366-368; the reference enables them together with the
        # cameras): device-side 15-step delay line, on request - the throughput configurations are proprioceptive only
        self.physics_state = self.delayed_physics_state = self._ps_ring = None
        if physics_state:
            self._ps_ring = z(native.PS_DELAY, native.PS_DIM, N)
            self.physics_state, self.delayed_physics_state = z(N, native.PS_DIM), z(N, native.PS_DIM)
            self.sim.bind_physics_state(self._ps_ring.data_ptr(), self.physics_state.data_ptr(), self.delayed_physics_state.data_ptr())
        nsub = int(round(task.control_timestep / PHYSICS_TIMESTEP))
        last = scenes.time_limit_last_step(time_limit, task.control_timestep, PHYSICS_TIMESTEP) if np.isfinite(time_limit) else 1 << 30
        self.last_step = last
        self.sim.configure(action_offset=[float(x) for x in task.calibration.homing_offsets], last_step=last,
                           n_substeps=nsub, solver_iterations=int(solver_iterations),
                           solver_tolerance=float(solver_tolerance), settle_max_substeps=int(settle_max_substeps),
                           terminate_on_success=int(task.terminate_episode), env_id_base=int(env_id_base),
                           solver={"newton": native.SOLVER_NEWTON, "pgs": native.SOLVER_PGS}[str(solver).lower()],
                           prefetch_resets=int(bool(prefetch_resets)))
        self.physics = _PhysicsView(self)
        self._physics = self.physics
        self._store = None
        # what a settled reset state depends on besides (env id, episode) and the mass scale: the key of the on-disk store
        self._settle_key = dict(
            blob_sha256=hashlib.sha256(blob).hexdigest(), seed=seed, env_id_base=int(env_id_base), n_envs=N,
            solver=str(solver).lower(), narrowphase=narrowphase, solver_iterations=int(solver_iterations), solver_tolerance=float(solver_tolerance),
            settle_max_substeps=int(settle_max_substeps), action_offset=[float(x) for x in task.calibration.homing_offsets])

    # ------------------------------------------------------------------ specs
    def action_spec(self) -> BoundedArray:
        return so100_action_spec(self.task.rotation_joint_limit)

    def observation_spec(self):
        spec = collections.OrderedDict()
        spec["commanded_joints_pos"] = Array((6,), np.float64, "commanded_joints_pos")
        spec["joints_pos"] = Array((6,), np.float64, "joints_pos")
        spec["joints_vel"] = Array((0,), np.float64, "joints_vel")
        with_state = self.physics_state is not None or (self.n_envs == 1 and self.task.image_observation_enabled)
        if with_state:
            spec["physics_state"] = Array((38,), np.float64, "physics_state")
        spec["undelayed_joints_pos"] = Array((6,), np.float64, "undelayed_joints_pos")
        spec["undelayed_joints_vel"] = Array((0,), np.float64, "undelayed_joints_vel")
        if with_state:
            spec["delayed_physics_state"] = Array((38,), np.float64, "delayed_physics_state")
        return spec

    # ------------------------------------------------------------------ stepping (tensors in, tensors out)
    def _stream(self):
        return self.torch.cuda.current_stream(self.device).cuda_stream

    def reset_all(self):
        self.sim.reset(None, self._stream())
        N = self.n_envs
        self.obs[:, 0:6] = self.qpos[0:6].t()
        self.obs[:, 6:12] = self.qpos[0:6].t()
        self.obs[:, 12:18] = self.ctrl.t()
        self.step_type.zero_()
        self.reward.zero_()
        self.discount.fill_(1.0)
        return self.obs

    def step_tensor(self, action):
        """action: float tensor [N, 6] on the device. Fills and returns (obs, reward, discount, step_type)."""
        torch = self.torch
        if action.shape != (self.n_envs, 6):
            raise ValueError(f"Expected 6 joint positions, got {tuple(action.shape)}")
        self._action.copy_(action)
        self.sim.step(self._action.data_ptr(), self.obs.data_ptr(), self.reward.data_ptr(), self.discount.data_ptr(),
                      self.step_type.data_ptr(), self._stream())
        return self.obs, self.reward, self.discount, self.step_type

    def _obs_dict(self):
        o = collections.OrderedDict()
        o["commanded_joints_pos"] = self.obs[:, 12:18]
        o["joints_pos"] = self.obs[:, 0:6]
        o["joints_vel"] = self.obs[:, 0:0]
        if self.physics_state is not None:
            o["physics_state"] = self.physics_state
        o["undelayed_joints_pos"] = self.obs[:, 6:12]
        o["undelayed_joints_vel"] = self.obs[:, 0:0]
        if self.physics_state is not None:
            o["delayed_physics_state"] = self.delayed_physics_state
        return o

    def reset(self) -> TimeStep:
        self.reset_all()
        return TimeStep(self.step_type, None, None, self._obs_dict())

    def step(self, action) -> TimeStep:
        torch = self.torch
        a = torch.as_tensor(action, dtype=torch.float32, device=self.device)
        if a.dim() == 1:
            a = a.unsqueeze(0)
        self.step_tensor(a)
        return TimeStep(self.step_type, self.reward, self.discount, self._obs_dict())

    def begin_episode(self):
        """Start an episode from the state currently in `qpos` / `qvel` (no placement, no settle)."""
        self.sim.begin_episode(self._stream())
        self.obs[:, 0:6] = self.qpos[0:6].t()
        self.obs[:, 6:12] = self.qpos[0:6].t()
        self.obs[:, 12:18] = self.ctrl.t()
        self.step_type.zero_()

    def set_mass_scale(self, scale):
        """scale: [2, N] (object, container) multipliers of the props' mass and inertia.  Flushes the reset prefetch
        (cached initial states were settled with the old masses)."""
        scale = self.torch.as_tensor(scale, dtype=self.torch.float32, device=self.device).reshape(2, self.n_envs)
        if not bool((scale > 0).all()):
            raise ValueError("mass scales must be positive")
        self.mass_scale.copy_(scale)
        if self._store is not None:
            self.set_settled_store(None)       # a settled-state store belongs to the old masses as well
        self.sim.configure()

    def set_reset_pool(self, qpos=None, qvel=None, ctrl=None):
        """Resets draw the initial state from this pool ([20, K], [18, K], [6, K] device tensors) instead of the
        reference's placement + settle; None restores the reference behaviour."""
        if qpos is None:
            self._pool = None
            self.sim.set_reset_pool(None, None, None, 0)
            return
        t = self.torch
        pool = tuple(t.as_tensor(a, dtype=t.float32, device=self.device).contiguous() for a in (qpos, qvel, ctrl))
        K = pool[0].shape[1]
        assert pool[0].shape == (20, K) and pool[1].shape == (18, K) and pool[2].shape == (6, K)
        self._pool = pool                      # keeps the tensors alive while the library reads them
        self.sim.set_reset_pool(pool[0].data_ptr(), pool[1].data_ptr(), pool[2].data_ptr(), K)

    # ------------------------------------------------------------------ settled-state store
    def settled_cache_key(self) -> dict:
        from . import build
        key = dict(self._settle_key)
        key["mass_scale_sha256"] = hashlib.sha256(self.mass_scale.detach().cpu().numpy().tobytes()).hexdigest()
        key["build"] = build.source_hash(mpr=self.narrowphase == "mpr")       # settled states are only bit-identical within one build of the kernels
        return key

    def compute_settled(self, n_episodes: int, first_episode: int = 0):
        """Placement + settle of episodes first_episode .. first_episode + n_episodes - 1 of every env, without touching
        the envs.  Returns device tensors (qpos [E,20,N], qvel [E,18,N], warmstart [E,18,N], flags [E,N])."""
        t, N, E = self.torch, self.n_envs, int(n_episodes)
        out = (t.zeros(E, 20, N, device=self.device), t.zeros(E, 18, N, device=self.device), t.zeros(E, 18, N, device=self.device),
               t.zeros(E, N, dtype=t.int32, device=self.device))
        self.sim.compute_settled(first_episode, E, *(a.data_ptr() for a in out), self._stream())
        return out

    def set_settled_store(self, tables=None, first_episode: int = 0):
        """Resets of the covered episodes copy their entry instead of settling (bit-identical); None detaches."""
        if tables is None:
            self._store = None
            self.sim.set_settled_store(None, None, None, None, 0, 0)
            return
        t, N = self.torch, self.n_envs
        q, v, w, f = tables
        q, v, w = (t.as_tensor(a, dtype=t.float32, device=self.device).contiguous() for a in (q, v, w))
        f = t.as_tensor(f, dtype=t.int32, device=self.device).contiguous()
        E = f.shape[0]
        assert q.shape == (E, 20, N) and v.shape == (E, 18, N) and w.shape == (E, 18, N) and f.shape == (E, N)
        self._store = (q, v, w, f)             # keeps the tensors alive while the library reads them
        self.sim.set_settled_store(q.data_ptr(), v.data_ptr(), w.data_ptr(), f.data_ptr(), int(first_episode), E)

    def save_settled_cache(self, path: str, n_episodes: int, first_episode: int = 0):
        """Computes the settled states of n_episodes episodes per env, writes them to `path` and attaches them."""
        from . import settled_cache
        tables = self.compute_settled(n_episodes, first_episode)
        names = [a[0] for a in settled_cache.ARRAYS]
        settled_cache.write(path, self.settled_cache_key(), first_episode, {n: a.cpu().numpy() for n, a in zip(names, tables)})
        self.set_settled_store(tables, first_episode)

    def load_settled_cache(self, path: str):
        """Attaches a file written by save_settled_cache(); refuses one computed for a different model, seed, shard,
        mass scale, solver setting or build (settled_cache.SettledCacheError names the differing fields)."""
        from . import settled_cache
        header, arrays = settled_cache.read(path, expect_key=self.settled_cache_key())
        self.set_settled_store(tuple(arrays[a[0]] for a in settled_cache.ARRAYS), header["first_episode"])
        return header

    def events(self, clear: bool = False) -> dict:
        """Counts since creation (or the last clear) of env-steps / env-resets that raised a flag: contact or candidate
        overflow, physics divergence (episode ended like a dm_control PhysicsError), rejected placement, unsettled
        reset.  See so101_get_events in include/so101.h."""
        self.sim.get_events(self._events.data_ptr(), clear, self._stream())
        v = self._events.cpu().tolist()
        return dict(zip(native.EVENT_NAMES, v))

    def episode_returns(self):
        out = self.torch.empty_like(self.ep_return)
        self.sim.get_returns(out.data_ptr(), self._stream())
        return out

    def diagnostics(self):
        d = self.torch.zeros(self.n_envs, native.DIAG_DIM, dtype=self.torch.int32, device=self.device)
        self.sim.get_diag(d.data_ptr(), self._stream())
        return d

    def synchronize(self):
        """Wait until everything this env was asked to do on the caller's current stream is done (steps, resets, reads).  Cheaper than
        `torch.cuda.synchronize()` right after a mass reset: a device-wide synchronise also waits for the library's background reset
        prefetch - settled initial states of FUTURE episodes, up to ~2 s of low-priority work on 256 wavefronts for 4096 envs - which no
        result of the calls made so far depends on (bench.py reports both clocks, `sustained.host_env_steps_per_s` / `env_steps_per_s`)."""
        self.torch.cuda.current_stream(self.device).synchronize()

    def close(self):
        self.sim.close()


class SingleEnvironment(BatchedEnvironment):
    """N = 1 facade with the reference's numpy observation dict (keys/order:
    examples/so101_rl_breakdown.ipynb:65; shapes :115-122).

    Seed compatibility: the reference's placements come from `np.random.RandomState(seed)` inside dm_control's
    PropPlacer.  With `seed_compatible=True` (default) this facade draws them from the same generator in the same
    order - object position `uniform(low, high)` (3 draws, so100_hand_over.py:37-41), object yaw (1, :42-49),
    container position (3 per attempt, at most 20 attempts against collisions, :51-55 / :216-221) - writes them into
    the state and lets the kernels settle (so101_settle) and start the episode.  An int seed or a RandomState therefore
    reproduces the reference's initial placements; the batched environments key a counter RNG by (seed, env id,
    episode) instead."""

    def __init__(self, task, seed_compatible: bool = True, random_state=None, **kw):
        self._seed_compatible = bool(seed_compatible)
        if isinstance(random_state, np.random.RandomState):
            self._np_random = random_state
        elif random_state is None:
            self._np_random = np.random.RandomState()
        else:
            self._np_random = np.random.RandomState(int(random_state))
        if self._seed_compatible:
            kw.setdefault("prefetch_resets", False)      # the host draws the placements: nothing the kernels could settle ahead of time
        kw.pop("physics_state", None)                    # N = 1 keeps the reference's rule (image_observation_enabled) and a host-side line
        super().__init__(task, n_envs=1, random_state=(random_state if not isinstance(random_state, np.random.RandomState) else 0), **kw)
        self._state_ring = collections.deque(maxlen=_PHYSICS_DELAY_STEPS)
        self._pending_first = False
        from .model import blob as blobfmt
        m = blobfmt.unpack(scenes.load_blob(task.object_name, "f64")[0])
        self._placer = dict(obj_lo=np.asarray(m["task_obj_pos_lo"], dtype=np.float64), obj_hi=np.asarray(m["task_obj_pos_hi"], dtype=np.float64),
                            yaw=np.asarray(m["task_obj_yaw"], dtype=np.float64), con_lo=np.asarray(m["task_con_pos_lo"], dtype=np.float64),
                            con_hi=np.asarray(m["task_con_pos_hi"], dtype=np.float64),
                            con_geoms=set(np.nonzero(np.asarray(m["geom_body"]) == int(np.asarray(m["task_container_body"]).ravel()[0]))[0].tolist()))
        self._dbg = self.torch.zeros(1, native.DEBUG_DIM, device=self.device)
        # a stream of its own: the library replays the step's launch chain as a captured HIP graph on any stream but the legacy null stream
        self._own = self.torch.cuda.Stream(device=self.device)

    def _join_caller(self):
        """Order the env's own stream behind the caller's current stream: the constructor's fills, `bind` / `configure`, and every
        inherited mutator (`set_mass_scale`, `set_reset_pool`, `begin_episode`, `step_tensor`, `compute_settled`) are enqueued on the stream
        that is current when they are called, and torch creates `_own` non-blocking - without this edge `env.set_mass_scale(t); env.reset()`
        would race on qpos / mass_scale / the record.  The other direction needs no event: `_fetch()` synchronises `_own` on the host
        before reset() / step() return, so whatever the caller enqueues afterwards starts after this env's work."""
        cur = self.torch.cuda.current_stream(self.device)
        if cur != self._own:
            self._own.wait_stream(cur)

    def _fetch(self):
        """the 256-byte state + output record of the last call -> pinned host memory, one copy, one synchronisation; returns the numpy view"""
        self._host.copy_(self._pack, non_blocking=True)
        self.torch.cuda.current_stream(self.device).synchronize()
        return self._host.numpy()

    # ---- reference-order placement + settle
    def _container_collides(self) -> bool:
        self.sim.debug_forward(self._dbg.data_ptr(), self._stream())
        r = self._dbg[0].cpu().numpy()
        D = native.DBG
        ncon = int(r[D["COUNTS"]])
        for k in range(ncon):
            g1, g2 = int(r[D["CON"] + 10 * k + 7]), int(r[D["CON"] + 10 * k + 8])
            if g1 in self._placer["con_geoms"] or g2 in self._placer["con_geoms"]:
                return True
        return False

    def _reset_seed_compatible(self):
        torch, P, rs = self.torch, self._placer, self._np_random
        opos = rs.uniform(P["obj_lo"], P["obj_hi"])                       # distributions.Uniform(low, high, single_sample=True)
        yaw = rs.uniform(P["yaw"][0], P["yaw"][1])                        # QuaternionFromAxisAngle(axis=z, angle=Uniform)
        q = np.zeros(20)
        q[6:9] = opos
        q[9:13] = [np.cos(0.5 * yaw), 0.0, 0.0, np.sin(0.5 * yaw)]
        q[16] = 1.0
        placed = False
        for _ in range(20):                                               # PropPlacer max_attempts_per_prop
            q[13:16] = rs.uniform(P["con_lo"], P["con_hi"])
            self.qpos.copy_(torch.as_tensor(q, dtype=torch.float32, device=self.device).unsqueeze(1))
            self.qvel.zero_()
            self.warm.zero_()
            if not self._container_collides():
                placed = True
                break
        if not placed:
            raise RuntimeError("Failed to place the container without collisions in 20 attempts (dm_control PropPlacer raises here too)")
        self.placements = dict(object_position=opos.copy(), object_yaw=float(yaw), container_position=q[13:16].copy())
        before = self.events()["settle_not_converged"]          # (cumulative counters stay what events() documents)
        # settle with the controls the episode starts with (ctrl = home + offsets), like the kernels' own reset
        home = np.asarray(scenes.SO100_HOME_CTRL, dtype=np.float64)
        self.ctrl.copy_(torch.as_tensor(home + np.asarray(self.task.calibration.homing_offsets, dtype=np.float64),
                                        dtype=torch.float32, device=self.device).unsqueeze(1))
        self.sim.settle(self._stream())
        self.begin_episode()
        if self.events()["settle_not_converged"] > before:
            import warnings
            warnings.warn("Failed to settle physics within the settle budget (dm_control warns likewise)")

    def _np_obs(self, first: bool, h=None):
        h = self._fetch() if h is None else h
        ob = h[38:56].astype(np.float64)
        o = collections.OrderedDict()
        o["commanded_joints_pos"] = ob[12:18].copy()
        o["joints_pos"] = ob[0:6].copy()
        o["joints_vel"] = np.zeros((0,), dtype=np.float64)
        if self.task.image_observation_enabled:
            state = h[0:38].astype(np.float64)
            if first:
                self._state_ring.clear()
                self._state_ring.extend([state] * _PHYSICS_DELAY_STEPS)   # INITIAL_VALUE padding
            delayed = self._state_ring[0]
            self._state_ring.append(state)
            o["physics_state"] = state
        o["undelayed_joints_pos"] = ob[6:12].copy()
        o["undelayed_joints_vel"] = np.zeros((0,), dtype=np.float64)
        if self.task.image_observation_enabled:
            o["delayed_physics_state"] = delayed
        return o

    def reset(self) -> TimeStep:
        self._join_caller()
        with self.torch.cuda.stream(self._own):
            if self._seed_compatible and self._pool is None:
                self._reset_seed_compatible()
            else:
                self.reset_all()
            self._pending_first = False
            return TimeStep(StepType.FIRST, None, None, self._np_obs(first=True))

    def step(self, action) -> TimeStep:
        a = np.asarray(action, dtype=np.float64).reshape(-1)
        if len(a) != 6:
            raise ValueError(f"Expected 6 joint positions, got {len(a)}")
        if self._pending_first:            # the step after LAST restarts the episode and reports FIRST (dm_control)
            return self.reset()
        self._join_caller()
        with self.torch.cuda.stream(self._own):
            self._act_host[0] = self.torch.from_numpy(a)
            self._action.copy_(self._act_host, non_blocking=True)
            self.sim.step(self._action.data_ptr(), self.obs.data_ptr(), self.reward.data_ptr(), self.discount.data_ptr(),
                          self.step_type.data_ptr(), self._stream())
            h = self._fetch()
        st = StepType(int(h[58:59].view(np.uint8)[0]))
        if st == StepType.LAST and self._seed_compatible and self._pool is None:
            self._pending_first = True     # the host draws the next placements; the kernels' own auto-reset is not used
        if st == StepType.FIRST:
            return TimeStep(st, None, None, self._np_obs(first=True, h=h))
        return TimeStep(st, float(h[56]), float(h[57]), self._np_obs(first=False, h=h))
