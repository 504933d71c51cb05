"""Import-path alias: `from so101_sim import task_suite` resolves to the MI355X package `so101_sim_amd`.

The reference's callers (`so101_sim/run_eval.py:22`, `scripts/so101_lerobot_wrapper.py:12`, `so101_rl.ipynb`,
`examples/so100_control_demo.py:10`) import the registry under this name; with this directory on the path
(the repo root) they run unchanged, without `so101_sim_amd.install_as_so101_sim()`.  Nothing is defined here:
the two modules re-export `so101_sim_amd` and `so101_sim_amd.task_suite`.
"""
import so101_sim_amd as _impl
from so101_sim_amd import __version__, install_as_so101_sim  # noqa: F401

__path__ = list(__path__)           # a regular package: `so101_sim.task_suite` is the module next to this file
AMD_BACKEND = _impl                 # the package doing the work
