"""Test helper: drives the C ABI (include/so101.h) with explicit state arrays.

backend="gpu": torch CUDA tensors + the in-tree libso101_hip.so (the product path, MI355X).
backend="emu": numpy arrays + tests/hostemu/_build/libso101_emu.so — the SAME kernel source compiled
               with g++ against a fake HIP runtime, one OS thread per lane.  Debug aid for CPU-only
               runs; it is never the thing whose parity is claimed.
"""
from __future__ import annotations

import os
import subprocess

import numpy as np

from so101_sim_amd import native

_HERE = os.path.dirname(os.path.abspath(__file__))
EMU_LIB = os.path.join(_HERE, "hostemu", "_build", "libso101_emu.so")
NQ, NV, NU = 20, 18, 6


def build_emu(mpr: bool = False):
    """mpr: the -DSO101_MPR build of the kernel source (MPR's own portal depth instead of the EPA expansion, so101_device.hpp)"""
    subprocess.check_call(["make", "-C", os.path.join(_HERE, "hostemu"), "-s"] + (["mpr"] if mpr else []))
    return EMU_LIB.replace("libso101_emu.so", "libso101_emu_mpr.so") if mpr else EMU_LIB


class ArraySim:
    def __init__(self, blob_f32: bytes, n_envs: int, backend: str = "gpu", seed: int = 0, mpr: bool = False, **cfg):
        self.N = n_envs
        self.backend = backend
        if backend == "gpu":
            import torch
            from so101_sim_amd import build as sbuild
            self.torch = torch
            self.dev = torch.device("cuda:0")
            # (pipeline 2 / 3: the experimental step paths live in libso101_hip_exp.so, built on demand like the MPR option)
            exp = int(cfg.get("pipeline", 1)) >= 2
            self.sim = native.Sim(blob_f32, n_envs, device=0, seed=seed, lib_path=sbuild.build(mpr=mpr, exp=exp) if (mpr or exp) else None)
            z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=self.dev)
            i32, u8 = torch.int32, torch.uint8
        else:
            self.sim = native.Sim(blob_f32, n_envs, device=0, seed=seed, lib_path=build_emu(mpr))
            z = lambda *s, dt=np.float32: np.zeros(s, dtype=dt)
            i32, u8 = np.int32, np.uint8
        N = n_envs
        self.qpos, self.qvel, self.ctrl, self.warm = z(NQ, N), z(NV, N), z(NU, N), z(NV, N)
        self.ring, self.ep_return = z(5, 6, N), z(N)
        self.step_count, self.episode = z(N, dt=i32), z(N, dt=i32)
        self.action, self.obs = z(N, 6), z(N, 18)
        self.reward_, self.discount, self.step_type = z(N), z(N), z(N, dt=u8)
        self.diag = z(N, native.DIAG_DIM, dt=i32)
        self.mass_scale = z(2, N)
        self._put(self.mass_scale, np.ones((2, N)))
        self._pool = None
        self.dbg = z(N, native.DEBUG_DIM)
        # free-body quaternions default to identity
        self.set_state(np.tile(np.concatenate([np.zeros(6), [0, 0, 0, 1, 0, 0, 0] * 2])[:, None], (1, N)))
        p = self.ptr
        self.sim.bind(p(self.qpos), p(self.qvel), p(self.ctrl), p(self.warm), p(self.ring), p(self.ep_return),
                      p(self.step_count), p(self.episode), p(self.mass_scale))
        if backend == "emu":
            # launches are synchronous in the emulator: a prefetch would settle every env after every call
            cfg.setdefault("prefetch_resets", 0)
            # the pipelined step is ~20 launches of 64 (k_order: 1024) OS threads per block here; the emulator runs the
            # fused step unless a test asks for the pipeline (test_pipelined_step_matches_fused does)
            cfg.setdefault("pipeline", 0)
        if cfg:
            self.sim.configure(**cfg)

    # -- helpers
    def ptr(self, a):
        return a.data_ptr() if self.backend == "gpu" else a.ctypes.data

    def _put(self, dst, src):
        src = np.asarray(src)
        if self.backend == "gpu":
            dst.copy_(self.torch.as_tensor(src.astype(np.float32) if dst.dtype == self.torch.float32 else src).to(self.dev).reshape(dst.shape))
        else:
            dst[...] = src.reshape(dst.shape)

    def _get(self, a):
        if self.backend == "gpu":
            self.torch.cuda.synchronize()
            return a.detach().cpu().numpy().copy()
        return a.copy()

    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream if self.backend == "gpu" else 0

    # -- state (arrays are [dim, N])
    def set_state(self, qpos=None, qvel=None, ctrl=None, warm=None):
        for dst, src in ((self.qpos, qpos), (self.qvel, qvel), (self.ctrl, ctrl), (self.warm, warm)):
            if src is not None:
                self._put(dst, src)

    def get_state(self):
        return self._get(self.qpos).astype(np.float64), self._get(self.qvel).astype(np.float64), self._get(self.warm).astype(np.float64)

    def configure(self, **kw):
        self.sim.configure(**kw)

    def set_mass_scale(self, scale):
        """scale [2, N]: per-env multiplier of the (object, container) mass and inertia"""
        self._put(self.mass_scale, np.asarray(scale, dtype=np.float32))
        self.sim.configure()

    def set_reset_pool(self, qpos, qvel, ctrl):
        """pool arrays [20, K], [18, K], [6, K]; None disables"""
        if qpos is None:
            self._pool = None
            self.sim.set_reset_pool(None, None, None, 0)
            return
        K = np.asarray(qpos).shape[1]
        if self.backend == "gpu":
            t = self.torch
            self._pool = tuple(t.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(self.dev) for a in (qpos, qvel, ctrl))
        else:
            self._pool = tuple(np.ascontiguousarray(a, dtype=np.float32) for a in (qpos, qvel, ctrl))
        self.sim.set_reset_pool(self.ptr(self._pool[0]), self.ptr(self._pool[1]), self.ptr(self._pool[2]), K)

    def compute_settled(self, n_episodes, first_episode=0):
        """-> numpy (qpos [E,20,N], qvel [E,18,N], warm [E,18,N], flags [E,N]) of so101_compute_settled"""
        E, N = n_episodes, self.N
        shapes = ((E, NQ, N), (E, NV, N), (E, NV, N), (E, N))
        if self.backend == "gpu":
            t = self.torch
            out = [t.zeros(*sh, dtype=(t.int32 if i == 3 else t.float32), device=self.dev) for i, sh in enumerate(shapes)]
        else:
            out = [np.zeros(sh, dtype=(np.int32 if i == 3 else np.float32)) for i, sh in enumerate(shapes)]
        self.sim.compute_settled(first_episode, E, *(self.ptr(a) for a in out), self.stream())
        return tuple(self._get(a) for a in out)

    def set_settled_store(self, tables, first_episode=0):
        if tables is None:
            self._store = None
            self.sim.set_settled_store(None, None, None, None, 0, 0)
            return
        dts = (np.float32, np.float32, np.float32, np.int32)
        if self.backend == "gpu":
            self._store = tuple(self.torch.as_tensor(np.ascontiguousarray(a, dtype=d)).to(self.dev) for a, d in zip(tables, dts))
        else:
            self._store = tuple(np.ascontiguousarray(a, dtype=d) for a, d in zip(tables, dts))
        self.sim.set_settled_store(*(self.ptr(a) for a in self._store), first_episode, self._store[3].shape[0])

    def get_events(self, clear=False):
        if self.backend == "gpu":
            ev = self.torch.zeros(native.NEVENTS, dtype=self.torch.int64, device=self.dev)
        else:
            ev = np.zeros(native.NEVENTS, dtype=np.int64)
        self.sim.get_events(self.ptr(ev), clear, self.stream())
        return dict(zip(native.EVENT_NAMES, [int(x) for x in self._get(ev)]))

    def physics(self, nsub=10, freeze_arm=False):
        self.sim.physics(nsub, freeze_arm, self.stream())

    def reset(self, mask=None):
        if mask is None:
            self.sim.reset(None, self.stream())
        else:
            m = np.asarray(mask, dtype=np.uint8)
            if self.backend == "gpu":
                mt = self.torch.as_tensor(m).to(self.dev)
                self.sim.reset(mt.data_ptr(), self.stream())
                self.torch.cuda.synchronize()
            else:
                self.sim.reset(m.ctypes.data, 0)

    def begin_episode(self):
        self.sim.begin_episode(self.stream())

    def step(self, action):
        self._put(self.action, np.asarray(action, dtype=np.float32))
        p = self.ptr
        self.sim.step(p(self.action), p(self.obs), p(self.reward_), p(self.discount), p(self.step_type), self.stream())
        return self._get(self.obs), self._get(self.reward_), self._get(self.discount), self._get(self.step_type)

    def reward(self):
        self.sim.reward(self.ptr(self.reward_), self.stream())
        return self._get(self.reward_)

    def get_diag(self):
        self.sim.get_diag(self.ptr(self.diag), self.stream())
        return self._get(self.diag)

    def debug_forward(self):
        self.sim.debug_forward(self.ptr(self.dbg), self.stream())
        d = self._get(self.dbg).astype(np.float64)
        D = native.DBG
        out = []
        for e in range(self.N):
            r = d[e]
            ncon = int(r[D["COUNTS"]])
            cons = [dict(pos=r[D["CON"] + 10 * k: D["CON"] + 10 * k + 3], normal=r[D["CON"] + 10 * k + 3: D["CON"] + 10 * k + 6],
                         dist=r[D["CON"] + 10 * k + 6], geom1=int(r[D["CON"] + 10 * k + 7]), geom2=int(r[D["CON"] + 10 * k + 8]),
                         dim=int(r[D["CON"] + 10 * k + 9]), force=r[D["FORCE"] + 6 * k: D["FORCE"] + 6 * k + 6]) for k in range(ncon)]
            out.append(dict(M=r[D["M"]:D["M"] + 36].reshape(6, 6), Minv=r[D["MINV"]:D["MINV"] + 36].reshape(6, 6),
                            bias=r[D["BIAS"]:D["BIAS"] + 6], qacc_smooth=r[D["SMOOTH"]:D["SMOOTH"] + 18],
                            qacc=r[D["QACC"]:D["QACC"] + 18], ncon=ncon, nrow=int(r[D["COUNTS"] + 1]),
                            iters=int(r[D["COUNTS"] + 2]), ncand=int(r[D["COUNTS"] + 3]), overflow=int(r[D["COUNTS"] + 4]),
                            xpos=r[D["XPOS"]:D["XPOS"] + 24].reshape(8, 3), contacts=cons,
                            rowf=r[D["ROWF"]:D["ROWF"] + 12], reward=r[D["REWARD"]]))
        return out


class TreeArraySim:
    """The general-tree engine (so101_tree_*, the ALOHA scenes) behind the same two backends."""

    def __init__(self, blob_f32: bytes, n_envs: int, backend: str = "gpu", mpr: bool = False):
        self.N, self.backend = n_envs, backend
        if backend == "gpu":
            import torch
            from so101_sim_amd import build as sbuild
            self.torch = torch
            self.dev = torch.device("cuda:0")
            self.sim = native.TreeSim(blob_f32, n_envs, device=0, lib_path=sbuild.build(mpr=True) if mpr else None)
            z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=self.dev)
            i32 = torch.int32
        else:
            self.sim = native.TreeSim(blob_f32, n_envs, device=0, lib_path=build_emu(mpr))
            z = lambda *s, dt=np.float32: np.zeros(s, dtype=dt)
            i32 = np.int32
        s = self.sim
        self.qpos, self.qvel, self.ctrl, self.warm = z(s.nq, n_envs), z(s.nv, n_envs), z(s.nu, n_envs), z(s.nv, n_envs)
        self.dbg = z(n_envs, s.debug_dim)
        self.diag = z(n_envs, 8, dt=i32)
        s.bind(self.ptr(self.qpos), self.ptr(self.qvel), self.ptr(self.ctrl), self.ptr(self.warm))

    def ptr(self, a):
        return a.data_ptr() if self.backend == "gpu" else a.ctypes.data

    def stream(self):
        return self.torch.cuda.current_stream().cuda_stream if self.backend == "gpu" else 0

    def _put(self, dst, src):
        src = np.asarray(src, dtype=np.float32).reshape(dst.shape)
        if self.backend == "gpu":
            dst.copy_(self.torch.from_numpy(src))
        else:
            dst[...] = src

    def _get(self, a):
        if self.backend == "gpu":
            self.torch.cuda.synchronize()
            return a.cpu().numpy()
        return a.copy()

    def set_state(self, qpos=None, qvel=None, ctrl=None, warm=None):
        for dst, src in ((self.qpos, qpos), (self.qvel, qvel), (self.ctrl, ctrl), (self.warm, warm)):
            if src is not None:
                self._put(dst, src)

    def get_state(self):
        return self._get(self.qpos).astype(np.float64), self._get(self.qvel).astype(np.float64), self._get(self.warm).astype(np.float64)

    def physics(self, nsub=10):
        self.sim.physics(nsub, self.stream())

    def get_diag(self):
        self.sim.get_diag(self.ptr(self.diag), self.stream())
        return self._get(self.diag)

    def debug_forward(self):
        self.sim.debug_forward(self.ptr(self.dbg), self.stream())
        d = self._get(self.dbg).astype(np.float64)
        D, nv, nb = self.sim.dbg, self.sim.nv, self.sim.nbody
        out = []
        for e in range(self.N):
            r = d[e]
            ncon = int(r[0])
            cons = [dict(pos=r[D["CON"] + 10 * k: D["CON"] + 10 * k + 3], normal=r[D["CON"] + 10 * k + 3: D["CON"] + 10 * k + 6],
                         dist=r[D["CON"] + 10 * k + 6], geom1=int(r[D["CON"] + 10 * k + 7]), geom2=int(r[D["CON"] + 10 * k + 8]),
                         dim=int(r[D["CON"] + 10 * k + 9]), fn=r[D["FORCE"] + k]) for k in range(ncon)]
            out.append(dict(ncon=ncon, nrow=int(r[1]), iters=int(r[2]), ncand=int(r[3]), flags=int(r[4]), nscalar=int(r[5]),
                            bias=r[D["BIAS"]:D["BIAS"] + nv], qacc_smooth=r[D["QSM"]:D["QSM"] + nv], qacc=r[D["QACC"]:D["QACC"] + nv],
                            xpos=r[D["XPOS"]:D["XPOS"] + 3 * nb].reshape(nb, 3), M=r[D["M"]:D["M"] + D["MSTRIDE"] ** 2].reshape(D["MSTRIDE"], D["MSTRIDE"])[:nv, :nv], contacts=cons))
        return out

    # -- env layer (hand-over scenes)
    def enable_env(self, **cfg):
        s, N = self.sim, self.N
        if self.backend == "gpu":
            t = self.torch
            z = lambda *sh, dt=t.float32: t.zeros(*sh, dtype=dt, device=self.dev)
            i32, u8 = t.int32, t.uint8
        else:
            z = lambda *sh, dt=np.float32: np.zeros(sh, dtype=dt)
            i32, u8 = np.int32, np.uint8
        jd = cfg.get("joints_delay_steps", -1); jd = 5 if jd < 0 else jd
        pd = cfg.get("physics_delay_steps", -1); pd = 15 if pd < 0 else pd
        self.ring_pos, self.ring_vel = z(max(jd, 1), s.nu, N), z(max(jd, 1), 16, N)
        self.ep_return, self.step_count, self.episode = z(N), z(N, dt=i32), z(N, dt=i32)
        self.action, self.obs = z(N, s.nu), z(N, s.obs_dim)
        self.reward_, self.discount, self.step_type = z(N), z(N), z(N, dt=u8)
        p = self.ptr
        s.bind_env(p(self.ring_pos), p(self.ring_vel), p(self.ep_return), p(self.step_count), p(self.episode))
        if cfg.pop("physics_state", False):
            D = s.nq + s.nv
            self.ps_ring, self.physics_state, self.delayed_physics_state = z(max(pd, 1), D, N), z(N, D), z(N, D)
            s.bind_physics_state(p(self.ps_ring), p(self.physics_state), p(self.delayed_physics_state))
        s.configure_env(**cfg)

    def reset(self):
        self.sim.reset(None, self.stream())

    def begin_episode(self):
        self.sim.begin_episode(self.stream())

    def step(self, action):
        self._put(self.action, action)
        p = self.ptr
        self.sim.step(p(self.action), p(self.obs), p(self.reward_), p(self.discount), p(self.step_type), self.stream())
        return self._get(self.obs).astype(np.float64), self._get(self.reward_).astype(np.float64), self._get(self.discount).astype(np.float64), self._get(self.step_type)
