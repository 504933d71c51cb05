"""ctypes wrapper of the CPU oracle — TEST INFRASTRUCTURE ONLY.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  The product
package (so101_sim_amd) never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libso101_oracle.so")


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("so101_oracle.cpp", "so101_oracle.hpp")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.exists(s) and os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class EnvCfg(C.Structure):
    _fields_ = [("offsets", C.c_double * 6), ("last_step", C.c_int), ("settle_max_substeps", C.c_int),
                ("seed", C.c_uint64), ("env_id", C.c_uint64)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        dp = C.POINTER(C.c_double)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_destroy.argtypes = [C.c_void_p]
        for n in ("orc_nq", "orc_nv", "orc_nu", "orc_ncon", "orc_nefc", "orc_solver_iter", "orc_env_step_count"):
            getattr(L, n).restype = C.c_int
            getattr(L, n).argtypes = [C.c_void_p]
        L.orc_set_solver.argtypes = [C.c_void_p, C.c_int, C.c_double]
        L.orc_set_collision.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_mass_scale.argtypes = [C.c_void_p, dp]
        L.orc_inject_contacts.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_set_solver_type.argtypes = [C.c_void_p, C.c_int]
        L.orc_ls_evals.restype = C.c_int
        L.orc_ls_evals.argtypes = [C.c_void_p]
        L.orc_set_state.argtypes = [C.c_void_p, dp, dp, dp]
        L.orc_get_state.argtypes = [C.c_void_p, dp, dp, dp]
        L.orc_set_ctrl.argtypes = [C.c_void_p, dp]
        L.orc_substeps.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_forward.argtypes = [C.c_void_p, C.c_int]
        L.orc_get_M.argtypes = [C.c_void_p, dp]
        L.orc_get_bias.argtypes = [C.c_void_p, dp]
        L.orc_get_qacc.argtypes = [C.c_void_p, dp, dp]
        L.orc_get_actuator_force.argtypes = [C.c_void_p, dp]
        L.orc_get_contact.argtypes = [C.c_void_p, C.c_int, dp]
        L.orc_get_efc_force.argtypes = [C.c_void_p, dp]
        L.orc_get_body_pose.argtypes = [C.c_void_p, C.c_int, dp, dp]
        L.orc_max_prop_qacc.restype = C.c_double
        L.orc_max_prop_qacc.argtypes = [C.c_void_p]
        L.orc_reward.restype = C.c_double
        L.orc_reward.argtypes = [C.c_void_p]
        L.orc_overlap_oobb.restype = C.c_int
        L.orc_overlap_oobb.argtypes = [dp, dp]
        L.orc_env_config.argtypes = [C.c_void_p, C.POINTER(EnvCfg)]
        L.orc_env_reset.argtypes = [C.c_void_p]
        L.orc_env_begin.argtypes = [C.c_void_p]
        L.orc_env_step.argtypes = [C.c_void_p, dp, dp, dp, dp, C.POINTER(C.c_int)]
        L.orc_env_obs.argtypes = [C.c_void_p, dp]
        L.orc_env_return.restype = C.c_double
        L.orc_env_return.argtypes = [C.c_void_p]
        L.orc_rng_uniform.restype = C.c_double
        L.orc_rng_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Oracle:
    """One fp64 environment."""

    def __init__(self, blob_f64: bytes):
        self.L = lib()
        self._blob = blob_f64
        self.h = self.L.orc_create(blob_f64, len(blob_f64))
        if not self.h:
            raise RuntimeError("oracle rejected the model blob (need the f64 blob, version match)")
        self.nq, self.nv, self.nu = self.L.orc_nq(self.h), self.L.orc_nv(self.h), self.L.orc_nu(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    # -- state
    def set_state(self, qpos=None, qvel=None, warm=None):
        a = [None if x is None else np.ascontiguousarray(x, dtype=np.float64) for x in (qpos, qvel, warm)]
        self.L.orc_set_state(self.h, *[None if x is None else _p(x) for x in a])

    def get_state(self):
        q, v, w = np.zeros(self.nq), np.zeros(self.nv), np.zeros(self.nv)
        self.L.orc_get_state(self.h, _p(q), _p(v), _p(w))
        return q, v, w

    def set_ctrl(self, ctrl):
        c = np.ascontiguousarray(ctrl, dtype=np.float64)
        self.L.orc_set_ctrl(self.h, _p(c))

    def set_solver(self, iterations=0, tolerance=-1.0):
        self.L.orc_set_solver(self.h, int(iterations), float(tolerance))

    def set_narrowphase(self, epa: bool):
        self.L.orc_set_narrowphase.argtypes = [C.c_void_p, C.c_int]
        self.L.orc_set_narrowphase(self.h, int(bool(epa)))

    def set_hull_multicontact(self, on: bool):
        self.L.orc_set_hull_multicontact.argtypes = [C.c_void_p, C.c_int]
        self.L.orc_set_hull_multicontact(self.h, int(bool(on)))

    def set_contact_capacity(self, capacity: int):
        """Mirror the kernels' contact capacity (native so101_max_contacts()): a substep with more contacts keeps ONE per touching geom pair.
        0 (the default) = no limit, like MuJoCo."""
        self.L.orc_set_contact_capacity.argtypes = [C.c_void_p, C.c_int]
        self.L.orc_set_contact_capacity(self.h, int(capacity))

    def contacts_reduced(self) -> int:
        self.L.orc_contacts_reduced.argtypes = [C.c_void_p]
        self.L.orc_contacts_reduced.restype = C.c_int
        return int(self.L.orc_contacts_reduced(self.h))

    def set_solver_type(self, newton: bool):
        self.L.orc_set_solver_type(self.h, int(bool(newton)))

    def set_mass_scale(self, scale):
        sc = np.ascontiguousarray(scale, dtype=np.float64)
        self.L.orc_set_mass_scale(self.h, _p(sc))

    def inject_contacts(self, contacts):
        """contacts: list of dicts (pos, normal, dist, geom1, geom2) used by forward() instead of the narrowphase; [] clears"""
        rows = np.array([[*c["pos"], *c["normal"], c["dist"], c["geom1"], c["geom2"]] for c in contacts], dtype=np.float64).reshape(-1, 9)
        self.L.orc_inject_contacts(self.h, len(contacts), _p(np.ascontiguousarray(rows)) if len(contacts) else None)

    def set_collision(self, enable: bool):
        self.L.orc_set_collision(self.h, int(enable))

    # -- physics
    def substeps(self, n=10, freeze_arm=False):
        return bool(self.L.orc_substeps(self.h, int(n), int(freeze_arm)))

    def forward(self, freeze_arm=False):
        self.L.orc_forward(self.h, int(freeze_arm))

    def M(self):
        m = np.zeros((self.nv, self.nv))
        self.L.orc_get_M(self.h, _p(m))
        return m

    def bias(self):
        b = np.zeros(self.nv)
        self.L.orc_get_bias(self.h, _p(b))
        return b

    def qacc(self):
        a, s = np.zeros(self.nv), np.zeros(self.nv)
        self.L.orc_get_qacc(self.h, _p(a), _p(s))
        return a, s

    def actuator_force(self):
        f = np.zeros(self.nu)
        self.L.orc_get_actuator_force(self.h, _p(f))
        return f

    def contacts(self):
        out = []
        for k in range(self.L.orc_ncon(self.h)):
            o = np.zeros(10)
            self.L.orc_get_contact(self.h, k, _p(o))
            out.append(dict(pos=o[0:3].copy(), normal=o[3:6].copy(), dist=o[6], geom1=int(o[7]), geom2=int(o[8]), dim=int(o[9])))
        return out

    def efc_force(self):
        f = np.zeros(max(self.L.orc_nefc(self.h), 1))
        self.L.orc_get_efc_force(self.h, _p(f))
        return f[: self.L.orc_nefc(self.h)]

    @property
    def nefc(self):
        return self.L.orc_nefc(self.h)

    @property
    def solver_iter(self):
        return self.L.orc_solver_iter(self.h)

    def body_pose(self, body):
        p, q = np.zeros(3), np.zeros(4)
        self.L.orc_get_body_pose(self.h, int(body), _p(p), _p(q))
        return p, q

    # -- task layer
    def reward(self):
        return self.L.orc_reward(self.h)

    def env_config(self, offsets=None, last_step=1 << 30, settle_max_substeps=1000, seed=0, env_id=0):
        cfg = EnvCfg()
        for k in range(6):
            cfg.offsets[k] = 0.0 if offsets is None else float(offsets[k])
        cfg.last_step, cfg.settle_max_substeps, cfg.seed, cfg.env_id = int(last_step), int(settle_max_substeps), int(seed), int(env_id)
        self.L.orc_env_config(self.h, C.byref(cfg))

    def env_reset(self):
        self.L.orc_env_reset(self.h)
        return self.env_obs()

    def env_begin(self):
        self.L.orc_env_begin(self.h)
        return self.env_obs()

    def env_obs(self):
        o = np.zeros(18)
        self.L.orc_env_obs(self.h, _p(o))
        return o

    def env_step(self, action):
        a = np.ascontiguousarray(action, dtype=np.float64)
        if a.shape != (6,):
            raise ValueError(f"Expected 6 joint positions, got {a.size}")
        o, r, d, st = np.zeros(18), C.c_double(), C.c_double(), C.c_int()
        self.L.orc_env_step(self.h, _p(a), _p(o), C.byref(r), C.byref(d), C.byref(st))
        return o, r.value, d.value, st.value


def overlap_oobb(box0, box1) -> bool:
    """box = (pos3, quat4, half3) concatenated; oobb_utils.overlap_oobb_oobb restated in C."""
    a = np.ascontiguousarray(np.concatenate([np.ravel(x) for x in box0]), dtype=np.float64)
    b = np.ascontiguousarray(np.concatenate([np.ravel(x) for x in box1]), dtype=np.float64)
    return bool(lib().orc_overlap_oobb(_p(a), _p(b)))


def rng_uniform(seed, env_id, episode, draw) -> float:
    return lib().orc_rng_uniform(int(seed), int(env_id), int(episode), int(draw))
