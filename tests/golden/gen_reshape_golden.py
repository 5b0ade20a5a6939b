#!/usr/bin/env python3
"""Golden vectors for ``ReshapeTokenization`` (reference ``preprocessing/tokenization.py:132-357``).

The reference module imports ``jax.numpy`` (not installed here), so it cannot be run.  What it
computes is pure index work, and the arithmetic is fully specified by two third-party calls the
reference makes with literal arguments:

* images: ``einops.rearrange(x, "b (h p1) (w p2) c -> b (h w) (p1 p2 c)", p1=ph, p2=pw)``
  (``tokenization.py:228-233``) and its inverse ``"b (h w) (p1 p2 c) -> b (h p1) (w p2) c"``
  (``:333-340``);
* audio: right zero-pad to a multiple of ``patch_length`` then ``reshape`` (``:236-263``).

This script evaluates exactly those calls with numpy + einops 0.8 (the ``einops`` the reference pins) and
writes inputs and expected outputs.  Integer-valued float32 inputs make every case bit-exact.

    python tests/golden/gen_reshape_golden.py     ->  tests/golden/reshape_tokenization_golden.npz
"""
from __future__ import annotations

import hashlib
import pathlib

import numpy as np
from einops import rearrange

OUT = pathlib.Path(__file__).resolve().parent / "reshape_tokenization_golden.npz"


def tok_image(x, ph, pw):
    if x.ndim == 2:
        h = w = int(np.sqrt(x.shape[1]))
        x = x.reshape(x.shape[0], h, w)
    if x.ndim == 3:
        x = x[..., None]
    return rearrange(x, "b (h p1) (w p2) c -> b (h w) (p1 p2 c)", p1=ph, p2=pw)


def detok_image(tokens, ph, pw, nh, nw):
    x = rearrange(tokens, "b (h w) (p1 p2 c) -> b (h p1) (w p2) c", h=nh, w=nw, p1=ph, p2=pw)
    return x[..., 0] if x.shape[3] == 1 else x


def tok_audio(x, L):
    if x.ndim == 3:
        x = x.reshape(x.shape[0], -1)
    T = x.shape[1]
    n = (T + L - 1) // L
    if T < n * L:
        x = np.concatenate([x, np.zeros((x.shape[0], n * L - T), dtype=x.dtype)], axis=1)
    return x.reshape(x.shape[0], n, L)


def main():
    rng = np.random.default_rng(20261004)
    out = {}
    # MNIST, flattened [B, 784], default 4x4 patches -> [B, 49, 16] (BASELINE configs #1/#2)
    x = rng.integers(-1000, 1000, size=(3, 784)).astype(np.float32)
    out["mnist_x"] = x
    out["mnist_p4_tokens"] = tok_image(x, 4, 4)
    out["mnist_p7_tokens"] = tok_image(x, 7, 7)
    out["mnist_p4_detok"] = detok_image(out["mnist_p4_tokens"], 4, 4, 7, 7)
    # rectangular patches (2, 14) on the 28x28 grid
    out["mnist_p2x14_tokens"] = tok_image(x, 2, 14)
    # colour image [B, 8, 12, 3] is not reachable through tokenize() (ndim 4 raises); the channel
    # interleave "(p1 p2 c)" is reachable on detokenize with patch_dim = p*p*C, so pin it there
    t3 = rng.integers(-50, 50, size=(2, 6, 48)).astype(np.float32)      # 6 = 2x3 patches of 4x4x3
    out["rgb_tokens"] = t3
    out["rgb_detok_img8x12"] = detok_image(t3, 4, 4, 2, 3)
    # audio, small with padding: T=1000 -> 8 patches of 128
    a = rng.integers(-3000, 3000, size=(2, 1000)).astype(np.float32)
    out["audio_small_x"] = a
    out["audio_small_tokens"] = tok_audio(a, 128)
    out["audio_small_L100_tokens"] = tok_audio(a, 100)
    # stereo [B, T, 2] -> channels flattened into time (interleaved)
    s = rng.integers(-3000, 3000, size=(2, 300, 2)).astype(np.float32)
    out["audio_stereo_x"] = s
    out["audio_stereo_tokens"] = tok_audio(s, 128)
    # the literal audio shape [B, 196608] -> [B, 1536, 128]: seed + digest + probes (the tensor itself is 1.5 MB)
    big = np.random.default_rng(7).integers(-30000, 30000, size=(2, 196608)).astype(np.float32)
    tb = tok_audio(big, 128)
    assert tb.shape == (2, 1536, 128)
    out["audio_literal_shape"] = np.array(tb.shape)
    out["audio_literal_sha256"] = np.frombuffer(hashlib.sha256(tb.tobytes()).digest(), dtype=np.uint8)
    out["audio_literal_probe_idx"] = np.array([[0, 0, 0], [0, 1, 0], [0, 1535, 127], [1, 700, 5], [1, 1, 1]])
    out["audio_literal_probe_val"] = np.array([tb[tuple(i)] for i in out["audio_literal_probe_idx"]], dtype=np.float32)
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
