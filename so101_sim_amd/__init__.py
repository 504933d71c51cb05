"""so101_sim_amd — MI355X (gfx950) batched step for the SO100/SO101 hand-over environments.

Public surface mirrors the reference package `so101_sim`:
    from so101_sim_amd import task_suite
    env = task_suite.create_task_env('SO100HandOverBanana', time_limit=30.0)           # N = 1, numpy obs
    envs = task_suite.create_task_env('SO100HandOverBanana', time_limit=10.0, n_envs=4096)  # torch tensors

`install_as_so101_sim()` registers this package under the reference's import path so notebooks and
harnesses that say `from so101_sim import task_suite` run unchanged.
"""
import os as _os
import sys as _sys

__version__ = "0.2.0"

# The pipelined step runs four launch chains + two service streams.  The HIP runtime maps streams onto
# GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that share a queue serialise, so ask for 8 - effective
# when this package is imported before HIP initialises (the library falls back to three chains otherwise).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")


def install_as_so101_sim():
    """Alias `so101_sim` / `so101_sim.task_suite` to this package."""
    from . import task_suite
    _sys.modules.setdefault("so101_sim", _sys.modules[__name__])
    _sys.modules.setdefault("so101_sim.task_suite", task_suite)
    return task_suite
