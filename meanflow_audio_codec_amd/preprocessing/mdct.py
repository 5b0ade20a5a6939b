"""MDCT / IMDCT on MI355X -- host mirror of ``preprocessing/mdct.py``.

Same public names, argument order, shapes and error behaviour as the reference
(``mdct`` :143-198, ``imdct`` :200-256, ``MDCTConfig`` :44-78), with torch
tensors instead of ``jnp`` arrays.  The arithmetic runs in the hand-written HIP
kernels of ``csrc/mdct.hip`` through the C ABI (``mfc_mdct_fwd`` /
``mfc_mdct_inv``).  ``use_fft_threshold`` is accepted for API compatibility and
ignored: the reference's FFT branch computes a different, non-invertible
transform (SURVEY defect 7); this build always evaluates the direct-path
definition (with a fast LDS-FFT algorithm for power-of-two windows).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch

from .. import _lib

DEFAULT_WINDOW_SIZE = 576
DEFAULT_FFT_THRESHOLD = 512


@dataclass
class MDCTConfig:
    """preprocessing/mdct.py:44-78."""
    window_size: int = DEFAULT_WINDOW_SIZE
    hop_size: int | None = None
    use_fft_threshold: int = DEFAULT_FFT_THRESHOLD

    def __post_init__(self) -> None:
        if self.window_size <= 0:
            raise ValueError(f"window_size must be positive, got {self.window_size}")
        if self.hop_size is not None and self.hop_size <= 0:
            raise ValueError(f"hop_size must be positive if provided, got {self.hop_size}")
        if self.use_fft_threshold <= 0:
            raise ValueError(f"use_fft_threshold must be positive, got {self.use_fft_threshold}")
        if self.hop_size is None:
            self.hop_size = self.window_size // 2


def _resolve_config(config, window_size, hop_size, use_fft_threshold):
    """preprocessing/mdct.py:437-469."""
    if config is not None:
        return config.window_size, config.hop_size, config.use_fft_threshold
    if window_size <= 0:
        raise ValueError(f"window_size must be positive, got {window_size}")
    if hop_size is not None and hop_size <= 0:
        raise ValueError(f"hop_size must be positive if provided, got {hop_size}")
    if use_fft_threshold <= 0:
        raise ValueError(f"use_fft_threshold must be positive, got {use_fft_threshold}")
    if hop_size is None:
        hop_size = window_size // 2
    return window_size, hop_size, use_fft_threshold


def num_frames(T: int, window_size: int, hop_size: int) -> int:
    """preprocessing/mdct.py:491."""
    return 1 if T < window_size else (T - window_size) // hop_size + 1


def mdct(x: torch.Tensor, window_size: int = DEFAULT_WINDOW_SIZE, hop_size: int | None = None,
         use_fft_threshold: int = DEFAULT_FFT_THRESHOLD, config: MDCTConfig | None = None) -> torch.Tensor:
    """Forward MDCT: ``(..., T) -> (..., n_frames, window_size)`` (mdct.py:143-198)."""
    if not isinstance(x, torch.Tensor):
        raise TypeError(f"Input must be a torch.Tensor, got {type(x)}")
    if x.ndim == 0:
        raise ValueError("Input must have at least 1 dimension")
    N, hop, _ = _resolve_config(config, window_size, hop_size, use_fft_threshold)
    _lib.require_cuda(x)
    lead = x.shape[:-1]
    T = x.shape[-1]
    x2 = x.reshape(-1, T).to(torch.float32).contiguous()
    B = x2.shape[0]
    nf = num_frames(T, N, hop)
    X = torch.empty((B, nf, N), dtype=torch.float32, device=x.device)
    if B > 0:
        rc = _lib.lib().mfc_mdct_fwd(x2.data_ptr(), B, T, T, N, hop, X.data_ptr(), _lib.stream_ptr())
        _lib.check(rc, "mfc_mdct_fwd")
    return X.reshape(*lead, nf, N)


def imdct(X: torch.Tensor, window_size: int = DEFAULT_WINDOW_SIZE, hop_size: int | None = None,
          use_fft_threshold: int = DEFAULT_FFT_THRESHOLD, config: MDCTConfig | None = None) -> torch.Tensor:
    """Inverse MDCT + overlap-add: ``(..., n_frames, window_size) -> (..., (n_frames-1)*hop + 2*window_size)``
    (mdct.py:200-256)."""
    if not isinstance(X, torch.Tensor):
        raise TypeError(f"Input must be a torch.Tensor, got {type(X)}")
    if X.ndim < 2:
        raise ValueError(f"Input must have at least 2 dimensions (n_frames, window_size), got shape {tuple(X.shape)}")
    N, hop, _ = _resolve_config(config, window_size, hop_size, use_fft_threshold)
    if X.shape[-1] != N:
        raise ValueError(f"last dimension ({X.shape[-1]}) must equal window_size ({N})")
    _lib.require_cuda(X)
    lead = X.shape[:-2]
    nf = X.shape[-2]
    X3 = X.reshape(-1, nf, N).to(torch.float32).contiguous()
    B = X3.shape[0]
    L = (nf - 1) * hop + 2 * N
    y = torch.empty((B, L), dtype=torch.float32, device=X.device)
    if B > 0:
        rc = _lib.lib().mfc_mdct_inv(X3.data_ptr(), B, nf, N, hop, y.data_ptr(), L, _lib.stream_ptr())
        _lib.check(rc, "mfc_mdct_inv")
    return y.reshape(*lead, L)
