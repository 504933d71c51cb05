"""Local look-alikes of the dm_env types every caller of the reference touches (dm_env is third
party and not installed here): StepType, TimeStep(.first/.mid/.last), specs.Array/BoundedArray.
Semantics follow SURVEY.md 8a-12: FIRST carries reward=None and discount=None
(examples/so101_rl_breakdown.ipynb:62-64)."""
from __future__ import annotations

import enum
from typing import Any, NamedTuple

import numpy as np


class StepType(enum.IntEnum):
    FIRST = 0
    MID = 1
    LAST = 2

    def first(self) -> bool:
        return self is StepType.FIRST

    def mid(self) -> bool:
        return self is StepType.MID

    def last(self) -> bool:
        return self is StepType.LAST


class TimeStep(NamedTuple):
    step_type: Any
    reward: Any
    discount: Any
    observation: Any

    def first(self) -> bool:
        return self.step_type == StepType.FIRST

    def mid(self) -> bool:
        return self.step_type == StepType.MID

    def last(self) -> bool:
        return self.step_type == StepType.LAST


class Array:
    def __init__(self, shape, dtype, name=None):
        self._shape = tuple(int(d) for d in shape)
        self._dtype = np.dtype(dtype)
        self._name = name

    shape = property(lambda self: self._shape)
    dtype = property(lambda self: self._dtype)
    name = property(lambda self: self._name)

    def __repr__(self):
        return f"Array(shape={self.shape}, dtype={self.dtype!r}, name={self.name!r})"

    def validate(self, value):
        value = np.asarray(value)
        if value.shape != self.shape:
            raise ValueError(f"Expected shape {self.shape} but found {value.shape}")
        return value

    def generate_value(self):
        return np.zeros(self.shape, self.dtype)


class BoundedArray(Array):
    def __init__(self, shape, dtype, minimum, maximum, name=None):
        super().__init__(shape, dtype, name)
        self._minimum = np.broadcast_to(np.asarray(minimum, dtype=self.dtype), self.shape).copy()
        self._maximum = np.broadcast_to(np.asarray(maximum, dtype=self.dtype), self.shape).copy()
        self._minimum.setflags(write=False)
        self._maximum.setflags(write=False)

    minimum = property(lambda self: self._minimum)
    maximum = property(lambda self: self._maximum)

    def __repr__(self):
        return (f"BoundedArray(shape={self.shape}, dtype={self.dtype!r}, name={self.name!r}, "
                f"minimum={self.minimum}, maximum={self.maximum})")
