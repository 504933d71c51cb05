"""Task registry and `create_task_env` with the reference's names, signature and error behaviour
(so101_sim/task_suite.py:43-100 registry, :103-155 factory), backed by the MI355X batched simulator.

All 22 registry keys are present.  Built: the SO100 hand-over tasks (SURVEY.md section 8 scope) and, on the general-tree
engine, the ALOHA hand-over tasks `HandOverBanana` / `HandOverPen` (section 8f-1, aloha.py); the other ALOHA tasks raise
NotImplementedError when constructed, not on import.
Extension over the reference: `n_envs` (default 1) and `device` select the batched GPU environment.
"""
from __future__ import annotations

import inspect
import types

import numpy as np

from . import aloha as _aloha
from . import env as _env

DEFAULT_CAMERAS = (
    "overhead_cam",
    "worms_eye_cam",
    "wrist_cam_left",
    "wrist_cam_right",
)

DEFAULT_CONTROL_TIMESTEP = 0.02


def _unbuilt(name):
    class _Unbuilt:
        __doc__ = f"{name}: ALOHA task outside the SO101 hot path (SURVEY.md 8f); not built."

        def __init__(self, **kwargs):
            raise NotImplementedError(
                f"{name} is an ALOHA bimanual task that is not built; this build covers the hand-over tasks (SO100 and ALOHA) and the Dining place-in-container tasks")
    _Unbuilt.__name__ = name
    return _Unbuilt


BlocksSpelling = _unbuilt("BlocksSpelling")
BowlOnRack = _unbuilt("BowlOnRack")
DesktopWrapHeadphone = _unbuilt("DesktopWrapHeadphone")
DiningPlaceInContainer = _aloha.DiningPlaceInContainerTask
DrawerOpen = _unbuilt("DrawerOpen")
HandOver = _aloha.HandOverTask
LaptopClose = _unbuilt("LaptopClose")
MarkerRemoveLid = _unbuilt("MarkerRemoveLid")
ToolsInCaddy = _unbuilt("ToolsInCaddy")
TowelFoldInHalf = _unbuilt("TowelFoldInHalf")
SO100HandOver = _env.SO100HandOverTask

_TOOLS = {f"ToolsPlace{tool_cc}In{side_cc}Compartment": (ToolsInCaddy, {"target_tool": tool, "target_compartment": side})
          for tool_cc, tool in (("Screwdriver", "screwdriver"), ("Magnifier", "magnifier"), ("CanOpener", "can_opener"), ("Scissors", "scissors"))
          for side_cc, side in (("Left", "left"), ("Right", "right"))}

TASK_FACTORIES = types.MappingProxyType({
    "BlocksSpelling": (BlocksSpelling, {}),
    "BowlOnRack": (BowlOnRack, {}),
    "DesktopWrapHeadphone": (DesktopWrapHeadphone, {}),
    "DiningPlaceBananaInBowl": (DiningPlaceInContainer, {"task_id": "banana"}),
    "DiningPlacePenInContainer": (DiningPlaceInContainer, {"task_id": "pen"}),
    "DiningPlaceMugOnPlate": (DiningPlaceInContainer, {"task_id": "mug"}),
    "DrawerOpen": (DrawerOpen, {}),
    "HandOverPen": (HandOver, {"object_name": "pen"}),
    "HandOverBanana": (HandOver, {"object_name": "banana"}),
    "LaptopClose": (LaptopClose, {}),
    "MarkerRemoveLid": (MarkerRemoveLid, {}),
    **_TOOLS,
    "TowelFoldInHalf": (TowelFoldInHalf, {}),
    # SO100 ARM tasks
    "SO100HandOverPen": (SO100HandOver, {"object_name": "pen"}),
    "SO100HandOverBanana": (SO100HandOver, {"object_name": "banana"}),
})


def create_task_env(
    task_name: str,
    time_limit: float,
    random_state: np.random.RandomState | int | None = None,
    control_timestep: float = DEFAULT_CONTROL_TIMESTEP,
    cameras: tuple[str, ...] = DEFAULT_CAMERAS,
    **kwargs,
):
    """Creates a task environment (same contract as the reference factory).

    Unknown `task_name` raises ValueError listing the names; kwargs that the task constructor does not
    name explicitly are silently dropped (task_suite.py:134-144) — for `SO100HandOver`, whose signature
    is `(object_name, reward_based_on_overlap=True, **kwargs)`, that is everything except those two.
    Batched extension: `n_envs`, `device`, `solver` ("newton" | "pgs"), `solver_iterations`, `solver_tolerance`,
    `settle_max_substeps`, `prefetch_resets`, `env_id_base`, `narrowphase` ("epa" default | "mpr"), `physics_state` (batched `physics_state` /
    `delayed_physics_state` observables, off by default: 38 + 38 floats per env and step) are consumed here and never
    reach the task.
    """
    if task_name not in TASK_FACTORIES:
        raise ValueError(
            f"Unknown task_name: {task_name}. Available tasks:"
            f" {list(TASK_FACTORIES.keys())}"
        )
    n_envs = int(kwargs.pop("n_envs", 1))
    env_kwargs = {k: kwargs.pop(k) for k in ("device", "solver_iterations", "solver_tolerance", "env_id_base", "settle_max_substeps", "solver",
                                                 "prefetch_resets", "physics_state", "seed_compatible", "narrowphase", "pipeline")
                  if k in kwargs}

    task_class, task_kwargs = TASK_FACTORIES[task_name]

    signature = inspect.signature(task_class.__init__)
    task_class_kwargs = set(signature.parameters.keys())
    task_class_kwargs.discard("self")
    kwargs = {k: v for k, v in kwargs.items() if k in task_class_kwargs}
    constructor_kwargs = {"control_timestep": control_timestep, "cameras": cameras, **task_kwargs}
    kwargs.update(constructor_kwargs)

    task_instance = task_class(**kwargs)
    if isinstance(task_instance, _aloha.HandOverTask):
        env_kwargs.pop("solver", None)                    # (a knob of the SO100 kernels only)
        return _aloha.AlohaEnvironment(task_instance, n_envs=n_envs, time_limit=time_limit, random_state=random_state, **env_kwargs)
    env_kwargs.pop("pipeline", None)                   # (a knob of the general-tree engine's env; the SO100 step path is chosen with env.sim.configure(pipeline=...))
    if n_envs == 1:
        return _env.SingleEnvironment(task_instance, time_limit=time_limit, random_state=random_state, **env_kwargs)
    env_kwargs.pop("seed_compatible", None)            # (single envs only: batches key the kernels' counter RNG)
    return _env.BatchedEnvironment(task_instance, n_envs=n_envs, time_limit=time_limit, random_state=random_state, **env_kwargs)
