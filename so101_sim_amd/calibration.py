"""The reference's calibration quirk (scripts/so101_calibration.py:13-88 with calibration/red_arm.json): per-joint
`homing_offset` values (raw encoder ticks) are ADDED to every action and to the home ctrl; the JSON is looked up relative
to the CURRENT WORKING DIRECTORY and silently ignored when absent or unreadable (offsets stay zero).  SURVEY.md section 9
item 1.  Only the offsets are consumed here (so101_config.action_offset)."""
from __future__ import annotations

import json

import numpy as np

JOINTS = ("shoulder_pan", "shoulder_lift", "elbow_flex", "wrist_flex", "wrist_roll", "gripper")


def homing_offsets(path: str = "calibration/red_arm.json") -> np.ndarray:
    """[6] offsets in the reference's joint order; zeros when the file is missing, unreadable or not JSON."""
    out = np.zeros(6)
    try:
        with open(path, "r") as f:
            data = json.load(f)
        for i, name in enumerate(JOINTS):
            out[i] = data.get(name, {}).get("homing_offset", 0)
    except (OSError, ValueError, AttributeError, TypeError):
        return np.zeros(6)
    return out


class SO101Calibration:
    """Name kept for callers of the reference class; `apply_calibration_to_action` adds the offsets (6 values or ValueError)."""

    def __init__(self, calibration_file: str = "calibration/red_arm.json"):
        self.homing_offsets = homing_offsets(calibration_file)

    def apply_calibration_to_action(self, action):
        if len(action) != 6:
            raise ValueError(f"Expected 6 joint positions, got {len(action)}")
        return np.asarray(action) + self.homing_offsets
