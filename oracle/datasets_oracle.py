"""CPU oracle for the data front end.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product path (``meanflow_audio_codec_amd``) never does.

Two restatements:

* ``resample_poly_f64`` -- the polyphase definition the device resampler implements,
  ``y[n] = sum_j x[j] h[n*down - j*up + half]`` in float64 with plain loops, and ``design_filter`` -- the published
  scipy design (``scipy.signal.resample_poly``: ``firwin(2*10*max(up,down)+1, 1/max(up,down), window=("kaiser", 5.0)) *
  up``).  The reference has no resampler (``datasets/audio.py:236-262`` keeps 44.1 kHz), so this row is pinned against
  scipy itself (``tests/test_datasets.py`` compares both restatements with ``scipy.signal.resample_poly``/``firwin``).
* ``frames_of`` / ``shuffle_order`` / ``batches_of`` -- the reference's iterator arithmetic
  (``datasets/audio.py:135-206, 264-277``: random left pad drawn from ``default_rng(seed).integers(0, frame_sz + 1)``
  once per file, right pad to a frame multiple, ``[n_frames, frame_sz, C]`` framing, swap-and-pop buffer shuffle with
  ``rng.integers(0, len - 1)``, batches of ``batch_size`` with an optional short tail) written as straight-line list
  code.  **Parity unpinned**: the reference module imports ``toolz`` and ``minimp3py``, neither of which is installed
  here, so it cannot be run to produce vectors and its tests hold none for this path.
"""
from __future__ import annotations

import math

import numpy as np


def design_filter(up: int, down: int, beta: float = 5.0) -> np.ndarray:
    g = math.gcd(up, down)
    up, down = up // g, down // g
    rate = max(up, down)
    half = 10 * rate
    m = np.arange(-half, half + 1, dtype=np.float64)
    fc = 1.0 / rate
    h = fc * np.sinc(fc * m) * np.kaiser(2 * half + 1, beta)
    h /= h.sum()
    return h * up


def resample_poly_f64(x: np.ndarray, up: int, down: int, h: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    T = x.shape[-1]
    nh = len(h)
    half = (nh - 1) // 2
    T_out = -(-T * up // down)
    y = np.zeros(x.shape[:-1] + (T_out,), dtype=np.float64)
    for n in range(T_out):
        p0 = n * down + half
        ja = max(0, -((nh - 1 - p0) // up))          # ceil((p0 - (nh-1)) / up)
        jb = min(T - 1, p0 // up)
        if jb < ja:
            continue
        j = np.arange(ja, jb + 1)
        y[..., n] = x[..., j] @ h[p0 - j * up]
    return y


def frames_of(files: list[np.ndarray], frame_sz: int, seed: int) -> list[np.ndarray]:
    """files: ``[C, n]`` arrays -> list of ``[frame_sz, C]`` frames in file order."""
    rng = np.random.default_rng(seed)
    out = []
    for a in files:
        pre = int(rng.integers(0, frame_sz + 1))
        post = (-(a.shape[-1] + pre)) % frame_sz
        p = np.concatenate([np.zeros((a.shape[0], pre), a.dtype), a, np.zeros((a.shape[0], post), a.dtype)], axis=1)
        for f in range(p.shape[1] // frame_sz):
            out.append(p[:, f * frame_sz:(f + 1) * frame_sz].T.copy())
    return out


def shuffle_order(n_items: int, buffer_size: int, seed: int) -> list[int]:
    """Indices of the items in the order the reference's buffer shuffle emits them."""
    rng = np.random.default_rng(seed)
    buf, out = [], []

    def pop():
        if len(buf) == 1:
            return buf.pop()
        i = int(rng.integers(0, len(buf) - 1))
        buf[i], buf[-1] = buf[-1], buf[i]
        return buf.pop()

    for k in range(n_items):
        buf.append(k)
        if len(buf) >= buffer_size:
            out.append(pop())
    while buf:
        out.append(pop())
    return out


def batches_of(items: list, batch_size: int, drop_last: bool) -> list[list]:
    out = [items[i:i + batch_size] for i in range(0, len(items), batch_size)]
    if out and len(out[-1]) < batch_size and drop_last:
        out.pop()
    return out
