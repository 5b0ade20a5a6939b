"""Timing / memory / parameter-count helpers -- host mirror of ``evaluators/performance.py``.

Same names and result dictionaries as the reference; the device side is HIP (``torch.cuda`` is the ROCm runtime
binding): timings are bracketed by a device synchronise, GPU memory comes from ``hipMemGetInfo``.
"""
from __future__ import annotations

import time
from contextlib import contextmanager
from typing import Any, Callable

import numpy as np
import torch

try:
    import psutil
except ImportError:          # optional, as in the reference (performance.py:17-20)
    psutil = None


def _sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()


class TrainingTimer:
    """``with TrainingTimer() as t: ...; t.elapsed()`` (``performance.py:23-51``)."""

    def __init__(self):
        self.start_time: float | None = None
        self.end_time: float | None = None

    def __enter__(self):
        _sync()
        self.end_time = None
        self.start_time = time.time()
        return self

    def __exit__(self, exc_type, exc, tb):
        _sync()
        self.end_time = time.time()
        return False

    def elapsed(self) -> float:
        if self.start_time is None:
            raise RuntimeError("Timer not started")
        return (time.time() if self.end_time is None else self.end_time) - self.start_time


def inference_time(fn: Callable, *args, num_warmup: int = 10, num_runs: int = 100, **kwargs) -> dict[str, float]:
    """Wall time per call of ``fn(*args, **kwargs)`` after ``num_warmup`` untimed calls; each timed call ends with a
    device synchronise (``performance.py:54-108``).  Keys: mean, std, min, max, total (seconds)."""
    for _ in range(num_warmup):
        fn(*args, **kwargs)
    _sync()
    laps = np.empty(num_runs, dtype=np.float64)
    for i in range(num_runs):
        t0 = time.perf_counter()
        fn(*args, **kwargs)
        _sync()
        laps[i] = time.perf_counter() - t0
    return {"mean": float(laps.mean()), "std": float(laps.std()), "min": float(laps.min()),
            "max": float(laps.max()), "total": float(laps.sum())}


def memory_usage() -> dict[str, Any]:
    """``gpu_memory_used_mb`` / ``gpu_memory_total_mb`` (device 0 of this process) and, with psutil, the host keys of
    ``performance.py:111-158``."""
    out: dict[str, Any] = {}
    if torch.cuda.is_available():
        free, total = torch.cuda.mem_get_info()
        out["gpu_memory_used_mb"] = (total - free) / 2 ** 20
        out["gpu_memory_total_mb"] = total / 2 ** 20
    if psutil is not None:
        try:
            out["cpu_memory_used_mb"] = psutil.Process().memory_info().rss / 2 ** 20
            vm = psutil.virtual_memory()
            out["cpu_memory_total_mb"] = vm.total / 2 ** 20
            out["cpu_memory_percent"] = vm.percent
        except Exception:
            pass
    return out


def count_parameters(params: Any) -> dict[str, Any]:
    """Total and per-leaf element counts of a parameter tree (``performance.py:161-199``).  Accepts this backend's
    flat ``{"blocks_0/input_proj1/kernel": tensor}`` dictionaries as well as nested ones; ``by_module`` is keyed by
    the ``/``-joined path of each leaf."""
    by_module: dict[str, int] = {}

    def walk(node, path):
        if isinstance(node, dict):
            for k, v in node.items():
                walk(v, f"{path}/{k}" if path else str(k))
        elif isinstance(node, (torch.Tensor, np.ndarray)):
            by_module[path] = by_module.get(path, 0) + int(node.numel() if isinstance(node, torch.Tensor) else node.size)

    walk(params, "")
    total = sum(by_module.values())
    return {"total": int(total), "total_millions": float(total / 1e6), "by_module": by_module, "trainable": int(total)}


@contextmanager
def memory_profiler():
    """``with memory_profiler() as p: ...`` then ``p["before"]``, ``p["after"]``, ``p["delta"]``
    (``performance.py:202-231``)."""
    prof: dict[str, Any] = {"before": memory_usage()}
    try:
        yield prof
    finally:
        after = memory_usage()
        prof["after"] = after
        before = prof["before"]
        delta = {}
        for k in set(before) | set(after):
            if k.endswith("_mb") or k.endswith("_percent"):
                name = k.replace("_mb", "_delta_mb").replace("_percent", "_delta_percent")
                delta[name] = after.get(k, 0) - before.get(k, 0)
        prof["delta"] = delta
