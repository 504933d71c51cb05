"""Generates tests/golden/lerobot_cases.json by EXECUTING the reference's own packaging code,
`SO101LeRobotWrapper._convert_to_lerobot_format` (/root/reference/scripts/so101_lerobot_wrapper.py:77-122), on
synthetic time steps.  The reference module imports `so101_sim.task_suite` (dm_control, not installable here) at import
time; the method under test never touches it, so an empty placeholder module is registered for that import and the
wrapper object is created without running its constructor (which would build the MuJoCo environment).
Run in the build container only (needs /root/reference); the JSON it writes is the committed fixture."""
import json
import os
import sys
import types
from collections import namedtuple

import numpy as np

REF = "/root/reference"
sys.modules.setdefault("so101_sim", types.ModuleType("so101_sim"))
sys.modules["so101_sim.task_suite"] = types.ModuleType("so101_sim.task_suite")
sys.modules["so101_sim"].task_suite = sys.modules["so101_sim.task_suite"]
sys.path.insert(0, os.path.join(REF, "scripts"))
import so101_lerobot_wrapper as ref          # noqa: E402

TimeStep = namedtuple("TimeStep", "step_type reward discount observation")


def main():
    rng = np.random.RandomState(0)
    w = object.__new__(ref.SO101LeRobotWrapper)
    w.device = "cpu"
    w.cameras = ()
    cases = []
    for k in range(12):
        joints = rng.uniform(-3.2, 3.2, 6) if k % 4 else np.zeros(6)
        action = None if k % 3 == 0 else rng.uniform(-3.2, 3.2, 6).astype(np.float64 if k % 2 else np.float32)
        w.frame_index = int(rng.randint(0, 2000)) if k else 0
        w.episode_index = int(rng.randint(0, 50))
        ts = TimeStep(1, 0.0, 1.0, {"joints_pos": joints, "undelayed_joints_pos": joints + 1.0})
        out = w._convert_to_lerobot_format(ts, action)
        cases.append(dict(
            joints_pos=joints.tolist(), action=None if action is None else [float(x) for x in action],
            action_dtype=None if action is None else str(action.dtype), frame_index=w.frame_index, episode_index=w.episode_index,
            expected={k2: (v if isinstance(v, str) else dict(dtype=str(v.dtype), shape=list(v.shape), value=v.double().flatten().tolist()))
                      for k2, v in out.items()}))
    out_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "lerobot_cases.json")
    json.dump(dict(source="scripts/so101_lerobot_wrapper.py:77-122 executed by scripts/make_golden_lerobot.py", cases=cases), open(out_path, "w"), indent=1)
    print("wrote", out_path, len(cases), "cases; keys", sorted(cases[0]["expected"]))


if __name__ == "__main__":
    main()
