"""`so101_sim.task_suite` = `so101_sim_amd.task_suite` (same objects, not copies): registry, factory, constants."""
from so101_sim_amd.task_suite import *  # noqa: F401,F403
from so101_sim_amd.task_suite import (DEFAULT_CAMERAS, DEFAULT_CONTROL_TIMESTEP, TASK_FACTORIES, create_task_env)  # noqa: F401
