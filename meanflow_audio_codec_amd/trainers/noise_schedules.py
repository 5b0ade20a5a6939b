"""Noise schedules -- host mirror of ``trainers/noise_schedules.py``.

The objects only carry the schedule constants; interpolation and target run fused with the noise
draw in ``mfc_flow_prepare``: z = (1-t) x0 + (noise_min + noise_max t) x1, target = noise_max x1 - x0
(LinearNoiseSchedule :52-88; UniformNoiseSchedule :91-115 is noise_min = 0, noise_max = 1).
"""
from __future__ import annotations

from abc import ABC

from .. import ops


class NoiseSchedule(ABC):
    noise_min: float
    noise_max: float

    def interpolate(self, x0, x1, t):
        z, _, _ = ops.flow_prepare(x0.contiguous(), t.contiguous(), x0.dtype, self.noise_min, self.noise_max,
                                   e=x1.contiguous())
        return z

    def compute_target(self, x0, x1):
        _, target, _ = ops.flow_prepare(x0.contiguous(), x0.new_zeros(x0.shape[0], 1), x0.dtype, self.noise_min,
                                        self.noise_max, e=x1.contiguous())
        return target


class LinearNoiseSchedule(NoiseSchedule):
    def __init__(self, noise_min: float = 0.001, noise_max: float = 0.999):
        self.noise_min = noise_min
        self.noise_max = noise_max


class UniformNoiseSchedule(NoiseSchedule):
    def __init__(self):
        self.noise_min = 0.0
        self.noise_max = 1.0
