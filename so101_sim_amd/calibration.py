"""Host mirror of the reference's calibration quirk (scripts/so101_calibration.py:13-88 with
calibration/red_arm.json): per-joint `homing_offset` values (raw encoder ticks) are ADDED to every
action and to the home ctrl; the JSON is looked up relative to the CURRENT WORKING DIRECTORY and
silently ignored when absent (offsets stay zero).  SURVEY.md section 9 item 1."""
from __future__ import annotations

import json
import os

import numpy as np

JOINT_MAPPING = {"shoulder_pan": 0, "shoulder_lift": 1, "elbow_flex": 2, "wrist_flex": 3, "wrist_roll": 4, "gripper": 5}


class SO101Calibration:
    def __init__(self, calibration_file: str = "calibration/red_arm.json"):
        self.calibration_file = calibration_file
        self.calibration_data: dict = {}
        self.joint_mapping = dict(JOINT_MAPPING)
        self.homing_offsets = np.zeros(6)
        self.load_calibration()

    def load_calibration(self) -> bool:
        try:
            if not os.path.exists(self.calibration_file):
                return False
            with open(self.calibration_file, "r") as f:
                self.calibration_data = json.load(f)
            for name, idx in self.joint_mapping.items():
                if name in self.calibration_data:
                    self.homing_offsets[idx] = self.calibration_data[name].get("homing_offset", 0)
            return True
        except Exception:
            return False

    def apply_calibration_to_position(self, joint_positions: np.ndarray) -> np.ndarray:
        if len(joint_positions) != 6:
            raise ValueError(f"Expected 6 joint positions, got {len(joint_positions)}")
        return joint_positions + self.homing_offsets

    def apply_calibration_to_action(self, action: np.ndarray) -> np.ndarray:
        return self.apply_calibration_to_position(action)

    def remove_calibration_from_position(self, calibrated_positions: np.ndarray) -> np.ndarray:
        if len(calibrated_positions) != 6:
            raise ValueError(f"Expected 6 joint positions, got {len(calibrated_positions)}")
        return calibrated_positions - self.homing_offsets
