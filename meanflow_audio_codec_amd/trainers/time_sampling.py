"""Time sampling -- host mirror of ``trainers/time_sampling.py``.

Draws come from the build's Philox stream keyed (seed, step, GLOBAL row) in ``mfc_sample_tr``
(JAX PRNG streams are not reproducible outside JAX; parity tests pass (t, r) explicitly).
"""
from __future__ import annotations

from abc import ABC, abstractmethod

import torch

from .. import ops


class PRNGKey:
    """(seed, counter) pair standing in for ``jax.random.PRNGKey``; ``next()`` advances it
    (the reference returns the key un-advanced from train_step -- defect 4, fixed here)."""
    __slots__ = ("seed", "counter")

    def __init__(self, seed: int, counter: int = 0):
        self.seed = int(seed)
        self.counter = int(counter)

    def next(self) -> "PRNGKey":
        return PRNGKey(self.seed, self.counter + 1)

    def __repr__(self):
        return f"PRNGKey(seed={self.seed}, counter={self.counter})"


class TimeSamplingStrategy(ABC):
    @abstractmethod
    def sample_time(self, key: PRNGKey, batch_size: int, dtype=torch.float32, row0: int = 0,
                    global_batch: int | None = None, device="cuda", row_stride: int = 1) -> torch.Tensor:
        """``row0`` / ``row_stride`` / ``global_batch``: local row i is global row ``row0 + i * row_stride`` of a
        batch of ``global_batch`` (data-parallel shards draw the rows of the global batch they own)."""
        ...


class UniformTimeSampling(TimeSamplingStrategy):
    """t ~ U[0,1] (time_sampling.py:39-49) -- host-side torch.rand is not used: Philox normal -> Phi."""
    def sample_time(self, key, batch_size, dtype=torch.float32, row0=0, global_batch=None, device="cuda", row_stride=1):
        Bg = global_batch or batch_size
        n = ops.randn(key.seed ^ 0x5EED, 0x7500 + (key.counter & 0xFFFF), 0, Bg, 1, device=device)
        n = n[row0:row0 + (batch_size - 1) * row_stride + 1:row_stride]
        return (0.5 * (1.0 + torch.erf(n / 2.0 ** 0.5))).to(dtype).contiguous()


class LogitNormalTimeSampling(TimeSamplingStrategy):
    """time_sampling.py:52-77."""
    def __init__(self, mean: float = -0.4, std: float = 1.0):
        self.mean = mean
        self.std = std

    def sample_time(self, key, batch_size, dtype=torch.float32, row0=0, global_batch=None, device="cuda", row_stride=1):
        t, _ = ops.sample_tr(key.seed, key.counter, row0, batch_size, global_batch or batch_size, self.mean,
                             self.std, 0.0, pair=False, device=device, row_stride=row_stride)
        return t


class MeanFlowTimeSampling(TimeSamplingStrategy):
    """time_sampling.py:79-135 / utils.sample_tr :36-45."""
    def __init__(self, mean: float = -0.4, std: float = 1.0, data_proportion: float = 0.5):
        self.mean = mean
        self.std = std
        self.data_proportion = data_proportion

    def sample_time(self, key, batch_size, dtype=torch.float32, row0=0, global_batch=None, device="cuda", row_stride=1):
        t, _ = ops.sample_tr(key.seed, key.counter, row0, batch_size, global_batch or batch_size, self.mean,
                             self.std, self.data_proportion, pair=False, device=device, row_stride=row_stride)
        return t

    def sample_time_pair(self, key, batch_size, dtype=torch.float32, row0=0, global_batch=None, device="cuda",
                         row_stride=1):
        return ops.sample_tr(key.seed, key.counter, row0, batch_size, global_batch or batch_size, self.mean,
                             self.std, self.data_proportion, pair=True, device=device, row_stride=row_stride)
