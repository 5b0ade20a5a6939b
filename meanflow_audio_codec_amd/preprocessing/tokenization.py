"""Tokenization strategies -- host mirror of ``preprocessing/tokenization.py``.

``MDCTTokenization`` (:46-129) runs the HIP MDCT kernels; ``ReshapeTokenization`` (:132-357) is pure
layout (einops-style patching) and stays a tensor view/copy -- there is no arithmetic in it.
"""
from __future__ import annotations

import math
from abc import ABC, abstractmethod

import torch

from .mdct import MDCTConfig, imdct, mdct, num_frames


class TokenizationStrategy(ABC):
    @abstractmethod
    def tokenize(self, x: torch.Tensor) -> torch.Tensor:
        """[B, ...] -> [B, n_tokens, token_dim]"""

    @abstractmethod
    def detokenize(self, tokens: torch.Tensor) -> torch.Tensor:
        """[B, n_tokens, token_dim] -> original format"""

    @abstractmethod
    def token_shape(self, original_dimension: int) -> tuple[int, int]:
        """(n_tokens, token_dim) for a flat [B, original_dimension] input, without launching anything."""


class MDCTTokenization(TokenizationStrategy):
    def __init__(self, window_size: int = 512, hop_size: int | None = None, config: MDCTConfig | None = None):
        self.config = config if config is not None else MDCTConfig(window_size=window_size, hop_size=hop_size)

    def tokenize(self, x):
        if x.ndim == 2:
            return mdct(x, config=self.config)
        if x.ndim == 3:
            # multi-channel: each channel separately, concatenated on the coefficient axis (:85-92)
            return torch.cat([mdct(x[:, :, c].contiguous(), config=self.config) for c in range(x.shape[2])], dim=-1)
        raise ValueError(f"Invalid input shape for MDCT: {tuple(x.shape)}")

    def detokenize(self, tokens):
        if tokens.ndim != 3:
            raise ValueError(f"Invalid tokens shape: {tuple(tokens.shape)}, expected [B, n_frames, ...]")
        N = self.config.window_size
        if tokens.shape[2] == N:
            return imdct(tokens, config=self.config)
        if tokens.shape[2] % N == 0:
            C = tokens.shape[2] // N
            return torch.stack([imdct(tokens[:, :, c * N:(c + 1) * N].contiguous(), config=self.config)
                                for c in range(C)], dim=-1)
        raise ValueError(f"Invalid tokens shape: {tuple(tokens.shape)}, token_dim ({tokens.shape[2]}) must be "
                         f"multiple of window_size ({N})")

    def token_shape(self, original_dimension):
        return num_frames(original_dimension, self.config.window_size, self.config.hop_size), self.config.window_size


class ReshapeTokenization(TokenizationStrategy):
    def __init__(self, patch_size=None, patch_length: int | None = None, image_size=None):
        self.patch_size = patch_size
        self.patch_length = patch_length
        self.image_size = image_size

    # -- dispatch (:159-187) -------------------------------------------------
    def _is_image(self, x):
        if x.ndim == 2:
            if self.image_size is not None or self.patch_size is not None:
                return True
            if self.patch_length is not None:
                return False
            return x.shape[1] == 784
        if x.ndim == 3:
            return x.shape[2] in (1, 3)
        raise ValueError(f"Invalid input shape for reshape tokenization: {tuple(x.shape)}")

    def tokenize(self, x):
        return self._tokenize_image(x) if self._is_image(x) else self._tokenize_audio(x)

    def _hw(self, total):
        if self.image_size is None:
            h = w = int(math.sqrt(total))
        else:
            h, w = (self.image_size, self.image_size) if isinstance(self.image_size, int) else self.image_size
        return h, w

    def _patch(self):
        if self.patch_size is None:
            return 4, 4
        if isinstance(self.patch_size, int):
            return self.patch_size, self.patch_size
        return tuple(self.patch_size)

    def _tokenize_image(self, x):
        if x.ndim == 2:
            h, w = self._hw(x.shape[1])
            x = x.reshape(x.shape[0], h, w)
        if x.ndim == 3:
            x = x[..., None]
        p1, p2 = self._patch()
        B, H, W, C = x.shape
        # "b (h p1) (w p2) c -> b (h w) (p1 p2 c)"
        x = x.reshape(B, H // p1, p1, W // p2, p2, C).permute(0, 1, 3, 2, 4, 5)
        return x.reshape(B, (H // p1) * (W // p2), p1 * p2 * C)

    def _tokenize_audio(self, x):
        if x.ndim == 3:
            x = x.reshape(x.shape[0], -1)
        L = 128 if self.patch_length is None else self.patch_length
        T = x.shape[1]
        n = (T + L - 1) // L
        if T < n * L:
            x = torch.cat([x, x.new_zeros(x.shape[0], n * L - T)], dim=1)
        return x.reshape(x.shape[0], n, L)

    def detokenize(self, tokens):
        patch_dim = tokens.shape[2]
        if self.patch_size is not None or self.image_size is not None:
            return self._detokenize_image(tokens)
        if self.patch_length is not None:
            return self._detokenize_audio(tokens)
        sq = int(math.sqrt(patch_dim))
        if sq * sq == patch_dim and sq <= 16:
            return self._detokenize_image(tokens)
        return self._detokenize_audio(tokens)

    def _detokenize_image(self, tokens):
        B, n_patches, patch_dim = tokens.shape
        if self.patch_size is None:
            sq = int(math.sqrt(patch_dim))
            if sq * sq == patch_dim:
                p1 = p2 = sq
                C = 1
            else:
                for p in (2, 4, 7, 8):
                    if patch_dim % (p * p) == 0:
                        p1 = p2 = p
                        C = patch_dim // (p * p)
                        break
                else:
                    p1 = p2 = 4
                    C = 1
        else:
            p1, p2 = self._patch()
            C = patch_dim // (p1 * p2)
        if self.image_size is None:
            # the reference leaves n_patches_per_side_{h,w} undefined here (NameError);
            # the evident intent (square grid) is implemented
            nh = nw = int(math.sqrt(n_patches))
        else:
            h, w = self._hw(0)
            nh, nw = h // p1, w // p2
        x = tokens.reshape(B, nh, nw, p1, p2, C).permute(0, 1, 3, 2, 4, 5).reshape(B, nh * p1, nw * p2, C)
        return x[..., 0] if C == 1 else x

    def _detokenize_audio(self, tokens):
        B, n, L = tokens.shape
        return tokens.reshape(B, n * L)

    def token_shape(self, original_dimension):
        image = (self.image_size is not None or self.patch_size is not None or
                 (self.patch_length is None and original_dimension == 784))
        if image:
            h, w = self._hw(original_dimension)
            p1, p2 = self._patch()
            return (h // p1) * (w // p2), p1 * p2
        L = 128 if self.patch_length is None else self.patch_length
        return (original_dimension + L - 1) // L, L
