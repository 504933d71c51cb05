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
def _hip_initialised() -> bool:
    torch = _sys.modules.get("torch")
    try:
        return bool(torch is not None and torch.cuda.is_initialized())
    except Exception:
        return False


if "GPU_MAX_HW_QUEUES" not in _os.environ:
    if _hip_initialised():
        # too late for the runtime to see it: tell the library to stay with three chains (it trusts the variable otherwise)
        import warnings as _warnings
        _warnings.warn("so101_sim_amd imported after HIP was initialised: GPU_MAX_HW_QUEUES cannot take effect any more, the "
                       "pipelined step uses three launch chains instead of four (export GPU_MAX_HW_QUEUES=8 before the first GPU call)")
        _os.environ["SO101_HW_QUEUES_EFFECTIVE"] = "4"
    else:
        _os.environ["GPU_MAX_HW_QUEUES"] = "16"     # (two handles in one process - the mixed suite - need 12 streams: 1.00 M with 8 queues, 1.12 M with 16)


def install_as_so101_sim():
    """Alias `so101_sim` / `so101_sim.task_suite` to this package."""
    from . import task_suite
    _sys.modules.setdefault("so101_sim", _sys.modules[__name__])
    _sys.modules.setdefault("so101_sim.task_suite", task_suite)
    return task_suite
