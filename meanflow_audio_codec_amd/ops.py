"""Thin torch-tensor wrappers over the C ABI (no arithmetic here)."""
from __future__ import annotations

import torch

from . import _lib
from ._lib import GEMM_ACCUM, GEMM_GELU, GEMM_TRANS_A, GEMM_TRANS_B  # noqa: F401


def gemm(A: torch.Tensor, B: torch.Tensor, *, trans_a=False, trans_b=False, bias=None, bias_rows=None,
         gelu=False, act_rows=None, alpha=1.0, residual=None, beta=1.0, out=None, accumulate=False,
         splitk=1, ws=None) -> torch.Tensor:
    """C = alpha*(op(A) op(B) + bias) + beta*residual through ``mfc_gemm``.

    A: [M,K] (or [K,M] if trans_a), B: [K,N] (or [N,K] if trans_b); 2-D, last
    dim contiguous.  bias is fp32 [N] applied to rows < bias_rows.
    """
    _lib.require_cuda(A, B)
    assert A.dim() == 2 and B.dim() == 2 and A.dtype == B.dtype
    assert A.stride(1) == 1 and B.stride(1) == 1
    M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
    K2, N = (B.shape[1], B.shape[0]) if trans_b else B.shape
    assert K == K2, (A.shape, B.shape, trans_a, trans_b)
    if out is None:
        out = torch.empty((M, N), dtype=A.dtype, device=A.device)
    assert out.shape == (M, N) and out.stride(1) == 1 and out.dtype == A.dtype
    flags = (GEMM_TRANS_A if trans_a else 0) | (GEMM_TRANS_B if trans_b else 0) \
        | (GEMM_ACCUM if accumulate else 0) | (GEMM_GELU if gelu else 0)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
    if bias_rows is None:
        bias_rows = M
    if act_rows is None:
        act_rows = M
    if (splitk > 1 or gelu) and ws is None:
        ws = torch.empty((M, N), dtype=torch.float32, device=A.device)
    if residual is not None:
        assert residual.shape == (M, N) and residual.dtype == A.dtype and residual.stride(1) == 1
    rc = _lib.lib().mfc_gemm(
        _lib.dtype_code(A.dtype), flags, M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0),
        out.data_ptr(), out.stride(0), _lib.ptr(bias), bias_rows, act_rows, float(alpha),
        _lib.ptr(residual), residual.stride(0) if residual is not None else 0, float(beta),
        int(splitk), _lib.ptr(ws), _lib.stream_ptr())
    _lib.check(rc, "mfc_gemm")
    return out
