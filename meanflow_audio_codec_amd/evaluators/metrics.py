"""Distribution and image metrics -- host mirror of ``evaluators/metrics.py``.

These are float64 host-side reductions over at most a few thousand small arrays (embeddings ``[N, D]``, 28x28 images);
they are not on the throughput path and stay in numpy exactly like the reference.  Values are pinned against the
reference module itself (``tests/golden/eval_metrics_golden.npz``, made by ``tests/golden/gen_metrics_golden.py``).

Three things the reference *does* rather than *says*, reproduced because results must be identical:

* ``_sqrtm_psd`` (``metrics.py:6-19``) multiplies ``V @ (sqrt(w)[None] * V.T)``, which scales the *columns* of ``V.T``:
  the product is ``V V^T diag(sqrt(w)) = diag(sqrt(w))`` -- the diagonal matrix of the square-rooted (ascending)
  eigenvalues, not the matrix square root.  ``frechet_distance`` inherits that (``frechet_distance(mu, S, mu, S)`` is
  not 0).  ``frechet_distance(..., reference_compatible=False)`` evaluates the actual Frechet distance.
* ``ssim(gaussian_weights=True)`` builds its window by Gaussian-filtering a *constant* array (``metrics.py:214-218``),
  which returns the constant: the window is the uniform ``1/win_size**2`` box for every ``sigma``.
* data-range inference (``metrics.py:126-133, 201-208``): any target inside ``[-1.1, 1.1]`` gets ``data_range = 2``;
  the ``[0, 1] -> 1`` branch behind it can never be taken.
"""
from __future__ import annotations

import numpy as np


def _sym_eigvals_sqrt(matrix: np.ndarray, floor: float) -> np.ndarray:
    sym = 0.5 * (matrix + matrix.T)
    return np.sqrt(np.maximum(np.linalg.eigvalsh(sym), floor))


def _sqrtm_psd(matrix: np.ndarray, eps: float = 1e-6, reference_compatible: bool = True) -> np.ndarray:
    """``metrics.py:6-19`` (see the module docstring for what it returns)."""
    if reference_compatible:
        return np.diag(_sym_eigvals_sqrt(matrix, eps))
    sym = 0.5 * (matrix + matrix.T)
    w, V = np.linalg.eigh(sym)
    return (V * np.sqrt(np.maximum(w, eps))) @ V.T


def frechet_distance(mu1: np.ndarray, sigma1: np.ndarray, mu2: np.ndarray, sigma2: np.ndarray,
                     reference_compatible: bool = True) -> float:
    """``|mu1 - mu2|^2 + Tr(S1 + S2 - 2 (S1^1/2 S2 S1^1/2)^1/2)`` with ``1e-6 I`` added to both covariances
    (``metrics.py:22-44``)."""
    mu1, mu2 = np.asarray(mu1, dtype=np.float64), np.asarray(mu2, dtype=np.float64)
    d = mu1.shape[0]
    jitter = 1e-6 * np.eye(d)
    s1 = np.asarray(sigma1, dtype=np.float64) + jitter
    s2 = np.asarray(sigma2, dtype=np.float64) + jitter
    root1 = _sqrtm_psd(s1, reference_compatible=reference_compatible)
    cross = _sqrtm_psd(root1 @ s2 @ root1, reference_compatible=reference_compatible)
    delta = mu1 - mu2
    return float(delta @ delta + np.trace(s1 + s2 - 2.0 * cross))


def kid_score(emb_real: np.ndarray, emb_fake: np.ndarray, subset_size: int = 100, num_subsets: int = 50,
              seed: int = 0) -> float:
    """Unbiased MMD^2 with the cubic kernel ``(x.y/d + 1)^3``, averaged over ``num_subsets`` random subsets
    (``metrics.py:47-99``).  The subset indices come from ``numpy.random.default_rng(seed)``, real first, then fake,
    once per subset -- the draw order is part of the result."""
    emb_real = np.asarray(emb_real)
    emb_fake = np.asarray(emb_fake)
    n_real, n_fake = emb_real.shape[0], emb_fake.shape[0]
    m = min(subset_size, n_real, n_fake)
    if m < 2:
        raise ValueError("subset_size must be >= 2 for KID computation")
    inv_d = 1.0 / float(emb_real.shape[1])
    gen = np.random.default_rng(seed)
    total = 0.0
    for _ in range(num_subsets):
        x = emb_real[gen.choice(n_real, m, replace=False)]
        y = emb_fake[gen.choice(n_fake, m, replace=False)]
        kxx = (x @ x.T * inv_d + 1.0) ** 3
        kyy = (y @ y.T * inv_d + 1.0) ** 3
        kxy = (x @ y.T * inv_d + 1.0) ** 3
        within = (kxx.sum() - np.trace(kxx) + kyy.sum() - np.trace(kyy)) / (m * (m - 1))
        total += within - 2.0 * kxy.mean()
    return float(total / num_subsets)


def _data_range(target: np.ndarray, data_range: float | None) -> float:
    if data_range is None:
        lo, hi = float(target.min()), float(target.max())
        data_range = 2.0 if (lo >= -1.1 and hi <= 1.1) else hi - lo
    if data_range <= 0:
        raise ValueError(f"Invalid data_range: {data_range}")
    return float(data_range)


def _pair(pred, target):
    pred = np.asarray(pred, dtype=np.float64)
    target = np.asarray(target, dtype=np.float64)
    if pred.shape != target.shape:
        raise ValueError(f"Shape mismatch: pred {pred.shape} vs target {target.shape}")
    return pred, target


def psnr(pred: np.ndarray, target: np.ndarray, data_range: float | None = None) -> float:
    """``20 log10(data_range / sqrt(mse))`` over the whole array, ``inf`` for identical inputs (``metrics.py:102-141``)."""
    pred, target = _pair(pred, target)
    mse = float(np.mean(np.square(pred - target)))
    if mse == 0:
        return float("inf")
    return float(20.0 * np.log10(_data_range(target, data_range) / np.sqrt(mse)))


def _box_mean(planes: np.ndarray, win: int) -> np.ndarray:
    """Mean over every fully-inside ``win x win`` window of the last two axes (``convolve2d(mode="valid")`` with the
    uniform window), as two 1-D sliding sums."""
    H, W = planes.shape[-2:]
    if H < win or W < win:
        raise ValueError(f"image {H}x{W} is smaller than the {win}x{win} SSIM window")
    rows = np.lib.stride_tricks.sliding_window_view(planes, win, axis=-2).sum(-1)
    both = np.lib.stride_tricks.sliding_window_view(rows, win, axis=-1).sum(-1)
    return both / float(win * win)


def ssim(pred: np.ndarray, target: np.ndarray, data_range: float | None = None, win_size: int = 11,
         gaussian_weights: bool = True, sigma: float = 1.5, k1: float = 0.01, k2: float = 0.03) -> float:
    """Mean SSIM over valid windows, channels and batch (``metrics.py:144-270``).  ``[H,W]``, ``[B,H,W]`` or
    ``[B,H,W,C]``; an even ``win_size`` is bumped to the next odd one; ``gaussian_weights``/``sigma`` are accepted and
    have no effect (module docstring)."""
    pred, target = _pair(pred, target)
    if pred.ndim == 2:
        pred, target = pred[None], target[None]
    win = win_size + 1 if win_size % 2 == 0 else win_size
    L = _data_range(target, data_range)
    c1, c2 = (k1 * L) ** 2, (k2 * L) ** 2
    if pred.ndim == 4:                                   # [B,H,W,C] -> [B,C,H,W]
        pred, target = np.moveaxis(pred, -1, 1), np.moveaxis(target, -1, 1)
    elif pred.ndim != 3:
        raise ValueError(f"expected [H,W], [B,H,W] or [B,H,W,C], got {pred.shape}")
    m1, m2 = _box_mean(pred, win), _box_mean(target, win)
    v1 = _box_mean(pred * pred, win) - m1 * m1
    v2 = _box_mean(target * target, win) - m2 * m2
    cov = _box_mean(pred * target, win) - m1 * m2
    smap = ((2.0 * m1 * m2 + c1) * (2.0 * cov + c2)) / ((m1 * m1 + m2 * m2 + c1) * (v1 + v2 + c2) + 1e-10)
    # mean over windows per channel, then channels, then batch == one flat mean (all groups have equal size)
    return float(smap.mean())
