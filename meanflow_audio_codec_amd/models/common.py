"""Shared host-side pieces of the velocity nets: parameter trees, initialisers, dtype policy.

Parameters are a flat ``dict[str, torch.Tensor]`` keyed by the Flax path joined with ``/``
(SURVEY Appendix B): ``Dense.kernel [in,out]``, ``Conv.kernel [kh,kw,in,out]``.  Masters are
always fp32; in bf16 mode every *big* ``kernel`` additionally has a bf16 working copy that the
HIP kernels read (``TrainState.work``).
"""
from __future__ import annotations

import math

import torch

from .. import ops


def lecun_normal_(t: torch.Tensor, fan_in: int, gen: torch.Generator) -> torch.Tensor:
    """flax ``lecun_normal``: truncated normal (+-2 sigma) with variance 1/fan_in."""
    std = math.sqrt(1.0 / fan_in) / 0.87962566103423978
    torch.nn.init.trunc_normal_(t, 0.0, 1.0, -2.0, 2.0, generator=gen)
    return t.mul_(std)


def init_from_shapes(shapes: dict, seed: int, device, big_threshold: int = 1 << 22) -> dict:
    """Initialise a flat ``name -> shape`` map.  kernels: lecun-normal; biases, GRN gamma/beta:
    zeros; layer_scale_gamma: 1e-6 (models/conv_flow.py:41-42,99-103).  Initialiser parity with
    Flax is unpinned by the reference."""
    gen = torch.Generator(device=device).manual_seed(seed)
    out = {}
    for name, shape in shapes.items():
        leaf = name.rsplit("/", 1)[-1]
        if leaf == "kernel":
            fan_in = math.prod(shape[:-1])
            t = torch.empty(shape, dtype=torch.float32, device=device)
            out[name] = lecun_normal_(t, fan_in, gen)
        elif leaf == "layer_scale_gamma":
            out[name] = torch.full(shape, 1e-6, dtype=torch.float32, device=device)
        elif leaf in ("query_tokens", "condition_tokens"):
            out[name] = torch.empty(shape, dtype=torch.float32, device=device).normal_(0, 0.02, generator=gen)
        else:
            out[name] = torch.zeros(shape, dtype=torch.float32, device=device)
    return out


def init_into(params: dict, shapes: dict, seed: int) -> dict:
    """``init_from_shapes`` into EXISTING tensors (same generator sequence, so the values equal a fresh
    ``init_from_shapes(shapes, seed, device)`` bit for bit): re-initialisation without a second parameter tree."""
    dev = next(iter(params.values())).device
    gen = torch.Generator(device=dev).manual_seed(seed)
    for name, shape in shapes.items():
        t = params[name]
        assert tuple(t.shape) == tuple(shape), name
        leaf = name.rsplit("/", 1)[-1]
        if leaf == "kernel":
            lecun_normal_(t, math.prod(shape[:-1]), gen)
        elif leaf == "layer_scale_gamma":
            t.fill_(1e-6)
        elif leaf in ("query_tokens", "condition_tokens"):
            t.normal_(0, 0.02, generator=gen)
        else:
            t.zero_()
    return params


def auto_splitk(M: int, N: int, K: int) -> int:
    """K slices of a product with few output tiles (a pure function of the shape: the slab sums are fixed-order, so
    results are reproducible).  One or two output tiles (the K = D / K = S products of a ConvFlow block): 512 slices =
    two workgroups per CU -- measured optimum at K = 392 704 and K = 6 270 016 (tools/sweep_splitk.py: 256..512 slices
    are 6-12 % faster than 1024, whose 64 MB of slabs per operand-GB start to show, and than 128, which starves HBM)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if 1024 <= K <= 8192 and tiles >= 16:
        # Mid-size products (the Mixer's token mixing and projections, BASELINE config #3): a power-of-two slice count,
        # slices at least 512 deep, chosen so that the workgroups fill the 256 CUs evenly -- at least two per CU and within
        # 10 % of a whole number of rounds (tools/sweep_splitk_mixer.py: 3072 x 2048 x 1024 runs 9 % faster in 2 slices than
        # in 1 = 384 workgroups = 1.5 rounds; 192 x 16384 x 1024, whose 64-row remainder is a launch of its own, 54 %).
        ntn, mfull, rem = (N + 127) // 128, M // 128, M % 128
        main = (mfull if (mfull >= 1 and 0 < rem <= 64) else (M + 127) // 128) * ntn     # workgroups of the main launch per slice
        # (K = 1024 with the CUs already covered stays whole: measured inside the training step, 3072 x 2048 x 1024 is 10 %
        # slower in two 512-deep slices although the isolated product is 9 % faster)
        depth = 512 if (K >= 2048 or main < 256) else 1024
        best, sk = 1, 1
        while K // sk >= depth:
            best = sk
            W = main * sk
            if W >= 512 and ((W + 255) // 256) * 256 <= 1.1 * W:
                break
            sk *= 2
        return best
    if K < 512 or tiles >= 256:
        return 1
    if K < 2048:
        # the small validation nets (BASELINE configs #2 / #3: a handful of output tiles, K ~ 1000): spread the tiles'
        # K range over the chip, at least 128 deep per slice -- one 128 x 128 fp32 tile per CU on 9 of 256 CUs otherwise
        return max(1, min(K // 128, (512 + tiles - 1) // tiles))
    target = 512 if tiles <= 2 else (1024 + tiles - 1) // tiles
    return max(1, min((K + 255) // 256, target))


def dense(x, w, b=None, *, bias_rows=None, out=None, alpha=1.0, residual=None, beta=1.0):
    """x [M,K] @ w [K,N] (+ b on the first bias_rows rows) through mfc_gemm."""
    M, K = x.shape
    N = w.shape[1]
    return ops.gemm(x, w, bias=b, bias_rows=bias_rows, out=out, alpha=alpha, residual=residual, beta=beta,
                    splitk=auto_splitk(M, N, K))


def dense_dx(dy, w, *, alpha=1.0, residual=None, beta=1.0, out=None):
    """dx [M,K] = dy [M,N] @ w[K,N]^T."""
    M, N = dy.shape
    K = w.shape[0]
    return ops.gemm(dy, w, trans_b=True, alpha=alpha, residual=residual, beta=beta, out=out,
                    splitk=auto_splitk(M, K, N))


def dense_dw(x, dy, *, alpha=1.0, out=None):
    """dw [K,N] = x [M,K]^T @ dy [M,N] (contraction over the M rows: split when the output has few tiles)."""
    return ops.gemm(x, dy, trans_a=True, alpha=alpha, out=out, splitk=auto_splitk(x.shape[1], dy.shape[1], x.shape[0]))
