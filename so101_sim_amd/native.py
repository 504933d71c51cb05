"""ctypes binding of libso101_hip.so (C ABI in include/so101.h).

The library is built in-tree by `__graft_entry__.build()` / `python -m so101_sim_amd.build`.  There is no
CPU fallback: if the shared object is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SO101_HIP_LIB") or os.path.join(_HERE, "csrc", "libso101_hip.so")      # (the variable: kernel experiments only)

SOLVER_PGS, SOLVER_NEWTON = 0, 1
OBS_DIM = 18
ACT_DIM = 6
RING_DEPTH = 5
DIAG_DIM = 8
PS_DIM, PS_DELAY = 38, 15
NEVENTS = 8
EVENT_NAMES = ("candidate_overflow", "contact_overflow", "arm_pool_overflow", "diverged", "placement_rejected",
               "settle_not_converged", "scheduler_abort", "contacts_reduced")
INFO = dict(graph_active=0, step_path=1, chains=2, hw_queues=3, scheduler_aborts=4, scratch_bytes=5)
DEBUG_DIM = 2048
MAXCON = 64

# debug_forward layout (csrc/so101_kernels.hpp DBG_*)
DBG = dict(M=0, MINV=36, BIAS=72, SMOOTH=78, QACC=96, COUNTS=114, XPOS=120, CON=144, FORCE=144 + 10 * MAXCON,
           ROWF=144 + 16 * MAXCON, REWARD=144 + 16 * MAXCON + 16)

EXPORTS = (
    "so101_version", "so101_max_contacts", "so101_create", "so101_destroy", "so101_default_config",
    "so101_configure", "so101_bind_state", "so101_bind_physics_state", "so101_set_reset_pool", "so101_compute_settled", "so101_set_settled_store", "so101_reset", "so101_settle", "so101_begin_episode", "so101_step", "so101_physics", "so101_reward",
    "so101_get_returns", "so101_get_diag", "so101_get_events", "so101_debug_forward", "so101_debug_candidates", "so101_debug_stages", "so101_get_info", "so101_debug_chain_stats", "so101_last_error",
    "so101_tree_create", "so101_tree_destroy", "so101_tree_dims", "so101_tree_last_plan", "so101_tree_bind_state", "so101_tree_configure", "so101_tree_physics",
    "so101_tree_debug_forward", "so101_tree_get_diag", "so101_tree_last_error", "so101_tree_obs_dim", "so101_tree_bind_env",
    "so101_tree_configure_env", "so101_tree_bind_physics_state", "so101_tree_reset", "so101_tree_step", "so101_tree_begin_episode", "so101_tree_settle", "so101_tree_compute_settled", "so101_tree_set_settled_store",
)


class Buffers(C.Structure):
    _fields_ = [("qpos", C.c_void_p), ("qvel", C.c_void_p), ("ctrl", C.c_void_p), ("warmstart", C.c_void_p),
                ("obs_ring", C.c_void_p), ("ep_return", C.c_void_p), ("step_count", C.c_void_p),
                ("episode", C.c_void_p), ("mass_scale", C.c_void_p)]


class Config(C.Structure):
    _fields_ = [("action_offset", C.c_float * ACT_DIM), ("last_step", C.c_int32), ("n_substeps", C.c_int32),
                ("solver_iterations", C.c_int32), ("solver_tolerance", C.c_float),
                ("settle_max_substeps", C.c_int32), ("terminate_on_success", C.c_int32),
                ("env_id_base", C.c_uint64), ("solver", C.c_int32), ("prefetch_resets", C.c_int32), ("pipeline", C.c_int32),
                ("groups", C.c_int32), ("use_graph", C.c_int32), ("chain_waves", C.c_int32)]


_libs: dict[str, C.CDLL] = {}


def load_library(path: str | None = None) -> C.CDLL:
    path = path or LIB_PATH
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback for the step path.")
    # PyTorch bundles its own HIP runtime; it must be the first one the process loads, otherwise this
    # library binds /opt/rocm's copy and the two runtimes do not see each other's device state.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp = C.c_void_p
    L.so101_version.restype = C.c_int
    L.so101_max_contacts.restype = C.c_int
    L.so101_create.restype = C.c_int
    L.so101_create.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_uint64, C.POINTER(vp)]
    L.so101_destroy.argtypes = [vp]
    L.so101_default_config.argtypes = [C.POINTER(Config)]
    L.so101_configure.argtypes = [vp, C.POINTER(Config)]
    L.so101_bind_state.argtypes = [vp, C.POINTER(Buffers)]
    L.so101_bind_physics_state.argtypes = [vp, vp, vp, vp]
    L.so101_set_reset_pool.argtypes = [vp, vp, vp, vp, C.c_int]
    L.so101_compute_settled.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp, vp, vp]
    L.so101_set_settled_store.argtypes = [vp, vp, vp, vp, vp, C.c_int, C.c_int]
    L.so101_reset.argtypes = [vp, vp, vp]
    L.so101_begin_episode.argtypes = [vp, vp]
    L.so101_settle.argtypes = [vp, vp]
    L.so101_step.argtypes = [vp, vp, vp, vp, vp, vp, vp]
    L.so101_physics.argtypes = [vp, C.c_int, C.c_int, vp]
    L.so101_reward.argtypes = [vp, vp, vp]
    L.so101_get_returns.argtypes = [vp, vp, vp]
    L.so101_get_diag.argtypes = [vp, vp, vp]
    L.so101_get_events.argtypes = [vp, vp, C.c_int, vp]
    L.so101_debug_forward.argtypes = [vp, vp, vp]
    L.so101_debug_candidates.argtypes = [vp, vp, vp, vp, vp, vp]
    L.so101_debug_stages.argtypes = [vp, vp, vp]
    L.so101_debug_chain_stats.argtypes = [vp, vp, C.c_int, vp]
    L.so101_get_info.restype = C.c_longlong
    L.so101_get_info.argtypes = [vp, C.c_int, vp]
    L.so101_last_error.restype = C.c_char_p
    L.so101_last_error.argtypes = [vp]
    _libs[path] = L
    return L


class Sim:
    """Thin handle wrapper. All array arguments are raw device addresses (ints)."""

    def __init__(self, blob_f32: bytes, n_envs: int, device: int = 0, seed: int = 0, lib_path: str | None = None):
        self.L = load_library(lib_path)
        self.n_envs = int(n_envs)
        h = C.c_void_p()
        rc = self.L.so101_create(blob_f32, len(blob_f32), self.n_envs, int(device), int(seed), C.byref(h))
        if rc != 0:
            msg = self.L.so101_last_error(None)
            raise RuntimeError(f"so101_create failed ({rc}): {msg.decode() if msg else '?'}")
        self.h = h
        self.cfg = Config()
        self.L.so101_default_config(C.byref(self.cfg))

    def close(self):
        if getattr(self, "h", None):
            self.L.so101_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.L.so101_last_error(self.h)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")

    def configure(self, **kw):
        for k, v in kw.items():
            if k == "action_offset":
                for i in range(ACT_DIM):
                    self.cfg.action_offset[i] = float(v[i])
            else:
                if not hasattr(self.cfg, k):
                    raise AttributeError(k)
                setattr(self.cfg, k, v)
        self._check(self.L.so101_configure(self.h, C.byref(self.cfg)), "so101_configure")

    def info(self, stream=0) -> dict:
        """Facts about the handle (so101_get_info): which step path ran, graph replay, scheduler aborts, scratch size."""
        return {k: int(self.L.so101_get_info(self.h, v, C.c_void_p(stream))) for k, v in INFO.items()}

    def chain_stats(self, clear=True, stream=0) -> dict:
        """Time accounting of the chained step's persistent kernel (so101_debug_chain_stats), seconds summed over wavefronts."""
        buf = (C.c_uint64 * 16)()
        self._check(self.L.so101_debug_chain_stats(self.h, C.cast(buf, C.c_void_p), int(clear), C.c_void_p(stream)), "so101_debug_chain_stats")
        k = ("t_pop", "t_idle", "t_narrow", "t_solve", "n_narrow", "n_solve", "n_idle", "waves", "t_life",
             "t_narrow_load", "t_narrow_pairs", "t_narrow_finish", "t_solve_load_gather", "t_solve_compute", "t_solve_store_broad", "t_solve_publish")
        d = {n: int(buf[i]) for i, n in enumerate(k)}
        for n in k:
            if not n.startswith("t_"):
                continue
            d[n] *= 1e-8
        return d

    def bind(self, qpos, qvel, ctrl, warmstart, obs_ring, ep_return, step_count, episode, mass_scale=None):
        b = Buffers(qpos, qvel, ctrl, warmstart, obs_ring, ep_return, step_count, episode, mass_scale)
        self._check(self.L.so101_bind_state(self.h, C.byref(b)), "so101_bind_state")

    def bind_physics_state(self, ring, physics_state, delayed):
        self._check(self.L.so101_bind_physics_state(self.h, *(C.c_void_p(x) if x else None for x in (ring, physics_state, delayed))),
                    "so101_bind_physics_state")

    def set_reset_pool(self, qpos, qvel, ctrl, pool_size: int):
        self._check(self.L.so101_set_reset_pool(self.h, qpos, qvel, ctrl, int(pool_size)), "so101_set_reset_pool")

    def compute_settled(self, first_episode: int, n_episodes: int, qpos, qvel, warm, flags, stream=0):
        self._check(self.L.so101_compute_settled(self.h, int(first_episode), int(n_episodes), qpos, qvel, warm, flags, stream),
                    "so101_compute_settled")

    def set_settled_store(self, qpos, qvel, warm, flags, first_episode: int, n_episodes: int):
        self._check(self.L.so101_set_settled_store(self.h, qpos, qvel, warm, flags, int(first_episode), int(n_episodes)),
                    "so101_set_settled_store")

    def reset(self, mask=None, stream=0):
        self._check(self.L.so101_reset(self.h, mask, stream), "so101_reset")

    def settle(self, stream=0):
        self._check(self.L.so101_settle(self.h, stream), "so101_settle")

    def begin_episode(self, stream=0):
        self._check(self.L.so101_begin_episode(self.h, stream), "so101_begin_episode")

    def step(self, action, obs, reward, discount, step_type, stream=0):
        self._check(self.L.so101_step(self.h, action, obs, reward, discount, step_type, stream), "so101_step")

    def physics(self, n_substeps: int, freeze_arm: bool = False, stream=0):
        self._check(self.L.so101_physics(self.h, int(n_substeps), int(freeze_arm), stream), "so101_physics")

    def reward(self, out, stream=0):
        self._check(self.L.so101_reward(self.h, out, stream), "so101_reward")

    def get_returns(self, out, stream=0):
        self._check(self.L.so101_get_returns(self.h, out, stream), "so101_get_returns")

    def get_diag(self, out, stream=0):
        self._check(self.L.so101_get_diag(self.h, out, stream), "so101_get_diag")

    def get_events(self, out, clear=False, stream=0):
        self._check(self.L.so101_get_events(self.h, out, int(bool(clear)), stream), "so101_get_events")

    def debug_candidates(self, ncand=None, cand=None, ticks=None, conres=None, stream=0):
        self._check(self.L.so101_debug_candidates(self.h, ncand, cand, ticks, conres, stream), "so101_debug_candidates")

    def debug_stages(self, out, stream=0):
        self._check(self.L.so101_debug_stages(self.h, out, stream), "so101_debug_stages")

    def debug_forward(self, out, stream=0):
        self._check(self.L.so101_debug_forward(self.h, out, stream), "so101_debug_forward")


class TreeConfig(C.Structure):
    _fields_ = [("n_substeps", C.c_int), ("last_step", C.c_int), ("settle_max_substeps", C.c_int), ("terminate_on_success", C.c_int),
                ("solver_iterations", C.c_int), ("solver_tolerance", C.c_float), ("seed", C.c_uint64), ("env_id_base", C.c_uint64),
                ("reward_mode", C.c_int), ("reward_requires_handover", C.c_int),
                ("joints_delay_steps", C.c_int), ("physics_delay_steps", C.c_int), ("prefetch_resets", C.c_int), ("pipeline", C.c_int)]


TREE_DBG = dict(COUNTS=0, BIAS=8, QSM=40, QACC=72, XPOS=104, M=200, CON=1224, FORCE=1864, MSTRIDE=32)     # the 32-dof build; TreeSim.dbg is the handle's own


class TreeSim:
    """Handle of the general-tree engine (include/so101.h, so101_tree_*): the ALOHA scenes.  Array arguments are raw device
    addresses (ints), state arrays are [dim][n_envs] float32."""

    def __init__(self, blob_f32: bytes, n_envs: int, device: int = 0, lib_path: str | None = None):
        self.L = load_library(lib_path)
        L = self.L
        L.so101_tree_create.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        L.so101_tree_destroy.argtypes = [C.c_void_p]
        L.so101_tree_destroy.restype = None
        L.so101_tree_dims.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.so101_tree_last_plan.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        L.so101_tree_bind_state.argtypes = [C.c_void_p] + [C.c_void_p] * 4
        L.so101_tree_configure.argtypes = [C.c_void_p, C.c_int, C.c_float]
        L.so101_tree_physics.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.so101_tree_debug_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.so101_tree_get_diag.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.so101_tree_last_error.argtypes = [C.c_void_p]
        L.so101_tree_last_error.restype = C.c_char_p
        L.so101_tree_obs_dim.argtypes = [C.c_void_p]
        L.so101_tree_bind_env.argtypes = [C.c_void_p] + [C.c_void_p] * 5
        L.so101_tree_configure_env.argtypes = [C.c_void_p, C.POINTER(TreeConfig)]
        L.so101_tree_reset.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.so101_tree_step.argtypes = [C.c_void_p] + [C.c_void_p] * 6
        L.so101_tree_begin_episode.argtypes = [C.c_void_p, C.c_void_p]
        L.so101_tree_settle.argtypes = [C.c_void_p, C.c_void_p]
        L.so101_tree_compute_settled.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 5
        L.so101_tree_set_settled_store.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_void_p] * 4
        self.n_envs = int(n_envs)
        h = C.c_void_p()
        rc = L.so101_tree_create(blob_f32, len(blob_f32), self.n_envs, int(device), C.byref(h))
        if rc != 0:
            msg = L.so101_tree_last_error(None)
            raise RuntimeError(f"so101_tree_create failed ({rc}): {msg.decode() if msg else '?'}")
        self.h = h
        d = (C.c_int * 16)()
        self._check(L.so101_tree_dims(h, d), "so101_tree_dims")
        self.nq, self.nv, self.nu, self.nbody, self.ngeom, self.debug_dim, self.max_contacts = list(d)[:7]
        # layout of so101_tree_debug_forward's row for this handle's build (32 or 64 dofs) and the build itself
        self.dbg = dict(COUNTS=0, BIAS=d[7], QSM=d[8], QACC=d[9], XPOS=d[10], M=d[11], CON=d[12], FORCE=d[13], MSTRIDE=d[14])
        self.build = int(d[15])
        self.obs_dim = int(L.so101_tree_obs_dim(h))
        self.cfg = TreeConfig(n_substeps=10, last_step=1 << 30, settle_max_substeps=1000, terminate_on_success=1, solver_iterations=0,
                              solver_tolerance=-1.0, seed=0, env_id_base=0, reward_mode=0, reward_requires_handover=0,
                              joints_delay_steps=-1, physics_delay_steps=-1, prefetch_resets=0, pipeline=0)

    def close(self):
        if getattr(self, "h", None):
            self.L.so101_tree_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            msg = self.L.so101_tree_last_error(self.h)
            raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")

    def bind(self, qpos, qvel, ctrl, warm):
        self._check(self.L.so101_tree_bind_state(self.h, qpos, qvel, ctrl, warm), "so101_tree_bind_state")

    def configure(self, solver_iterations: int = 0, solver_tolerance: float = -1.0):
        self._check(self.L.so101_tree_configure(self.h, int(solver_iterations), float(solver_tolerance)), "so101_tree_configure")

    def physics(self, n_substeps: int, stream: int = 0):
        self._check(self.L.so101_tree_physics(self.h, int(n_substeps), stream), "so101_tree_physics")

    def debug_forward(self, out, stream: int = 0):
        self._check(self.L.so101_tree_debug_forward(self.h, out, stream), "so101_tree_debug_forward")

    def get_diag(self, out, stream: int = 0):
        self._check(self.L.so101_tree_get_diag(self.h, out, stream), "so101_tree_get_diag")

    def bind_env(self, ring_pos, ring_vel, ep_return, step_count, episode):
        self._check(self.L.so101_tree_bind_env(self.h, ring_pos, ring_vel, ep_return, step_count, episode), "so101_tree_bind_env")

    def bind_physics_state(self, ring, physics_state, delayed):
        self.L.so101_tree_bind_physics_state.argtypes = [C.c_void_p] * 4
        self._check(self.L.so101_tree_bind_physics_state(self.h, *(C.c_void_p(x) if x else None for x in (ring, physics_state, delayed))),
                    "so101_tree_bind_physics_state")

    def configure_env(self, **kw):
        for k, v in kw.items():
            if not hasattr(self.cfg, k):
                raise TypeError(f"unknown config field {k}")
            setattr(self.cfg, k, v)
        self._check(self.L.so101_tree_configure_env(self.h, C.byref(self.cfg)), "so101_tree_configure_env")

    def reset(self, mask=None, stream: int = 0):
        self._check(self.L.so101_tree_reset(self.h, mask, stream), "so101_tree_reset")

    def last_plan(self):
        """(env slices, kernel launches, memsets, path) of the last so101_tree_step as the library enqueued it; path 0 = no step yet"""
        d = (C.c_int * 4)()
        self._check(self.L.so101_tree_last_plan(self.h, d), "so101_tree_last_plan")
        return tuple(d)

    def step(self, action, obs, reward, discount, step_type, stream: int = 0):
        self._check(self.L.so101_tree_step(self.h, action, obs, reward, discount, step_type, stream), "so101_tree_step")

    def begin_episode(self, stream: int = 0):
        self._check(self.L.so101_tree_begin_episode(self.h, stream), "so101_tree_begin_episode")

    def settle(self, stream: int = 0):
        self._check(self.L.so101_tree_settle(self.h, stream), "so101_tree_settle")

    def compute_settled(self, first_episode, count, qpos, qvel, warm, flags, stream: int = 0):
        self._check(self.L.so101_tree_compute_settled(self.h, int(first_episode), int(count), qpos, qvel, warm, flags, stream), "so101_tree_compute_settled")

    def set_settled_store(self, first_episode, count, qpos, qvel, warm, flags):
        self._check(self.L.so101_tree_set_settled_store(self.h, int(first_episode), int(count), qpos, qvel, warm, flags), "so101_tree_set_settled_store")
