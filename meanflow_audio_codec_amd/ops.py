"""Thin torch-tensor wrappers over the C ABI (no arithmetic here)."""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import GEMM_ACCUM, GEMM_GELU, GEMM_LN16, GEMM_LN16T, GEMM_TRANS_A, GEMM_TRANS_B  # noqa: F401


def gemm(A: torch.Tensor, B: torch.Tensor, *, trans_a=False, trans_b=False, bias=None, bias_rows=None,
         gelu=False, act_rows=None, alpha=1.0, residual=None, beta=1.0, out=None, accumulate=False,
         splitk=1, ws=None, ln_rstd=None, ln_tangent=False) -> torch.Tensor:
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
        | (GEMM_ACCUM if accumulate else 0) | (GEMM_GELU if gelu else 0) | (GEMM_LN16 if ln_rstd is not None else 0) \
        | (GEMM_LN16T if (ln_rstd is not None and ln_tangent) else 0)
    if ln_rstd is not None:
        assert ln_rstd.dtype == torch.float32 and ln_rstd.is_contiguous() and N % 16 == 0
        assert ln_rstd.numel() >= (M if bias_rows is None else bias_rows) * (N // 16)
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == N
    if bias_rows is None:
        bias_rows = M
    if act_rows is None:
        act_rows = M
    if (splitk > 1 or gelu):
        need = int(_lib.lib().mfc_gemm_ws_elems(flags, M, N, K, int(splitk)))   # one [M, N] slab per K slice
        if ws is None or ws.numel() < need:
            ws = torch.empty(max(need, 1), dtype=torch.float32, device=A.device)
        assert ws.dtype == torch.float32 and ws.is_contiguous()
    if residual is not None:
        assert residual.shape == (M, N) and residual.dtype == A.dtype and residual.stride(1) == 1
    rc = _lib.lib().mfc_gemm(
        _lib.dtype_code(A.dtype), flags, M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0),
        out.data_ptr(), out.stride(0), _lib.ptr(bias), bias_rows, act_rows, float(alpha),
        _lib.ptr(residual), residual.stride(0) if residual is not None else 0, float(beta),
        int(splitk), _lib.ptr(ws), _lib.ptr(ln_rstd), _lib.stream_ptr())
    _lib.check(rc, "mfc_gemm")
    return out


def gemm_adamw(A, B, *, p, m, v, p_bf16, lr, wd, step, trans_a=False, trans_b=False, grad_scale=1.0, b1=0.9,
               b2=0.999, eps=1e-8, colsum=None, colsum_scale=1.0) -> None:
    """Weight gradient op(A) op(B) fused with the AdamW update of that weight (``mfc_gemm_adamw``): updates the fp32
    master ``p``, the moments ``m``/``v`` and the bf16 working copy ``p_bf16`` in place; bit-identical to
    ``gemm(..., out=g_bf16)`` followed by ``adamw(p, g_bf16, ...)``.  ``colsum`` (fp32 [N]): overwritten with
    ``colsum_scale`` x the column sums of ``B`` (the bias gradient when ``B`` = dY), computed from the operand tiles the
    product stages anyway."""
    _lib.require_cuda(A, B)
    assert A.dtype == B.dtype == torch.bfloat16 and A.dim() == 2 and B.dim() == 2 and A.stride(1) == 1 and B.stride(1) == 1
    M, K = (A.shape[1], A.shape[0]) if trans_a else A.shape
    K2, N = (B.shape[1], B.shape[0]) if trans_b else B.shape
    assert K == K2, (A.shape, B.shape, trans_a, trans_b)
    for t_ in (p, m, v):
        assert t_.dtype == torch.float32 and t_.is_contiguous() and t_.numel() == M * N
    assert p_bf16.dtype == torch.bfloat16 and p_bf16.is_contiguous() and p_bf16.numel() == M * N
    flags = (GEMM_TRANS_A if trans_a else 0) | (GEMM_TRANS_B if trans_b else 0)
    if colsum is not None:
        assert colsum.dtype == torch.float32 and colsum.is_contiguous() and colsum.numel() == N and not trans_b
    rc = _lib.lib().mfc_gemm_adamw(flags, M, N, K, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), float(grad_scale),
                                   p.data_ptr(), m.data_ptr(), v.data_ptr(), p_bf16.data_ptr(), float(lr), float(b1),
                                   float(b2), float(eps), float(wd), int(step), _lib.ptr(colsum), float(colsum_scale),
                                   _lib.stream_ptr())
    _lib.check(rc, "mfc_gemm_adamw")


# ---------------------------------------------------------------------------
# ConvNeXt block interior
# ---------------------------------------------------------------------------
import ctypes  # noqa: E402

_CNX_FIELDS = ("conv_w", "conv_b", "exp_w", "exp_b", "grn_gamma", "grn_beta", "con_w", "con_b", "ls")


def _cnx_struct(tensors: dict) -> _lib.CnxParams:
    s = _lib.CnxParams()
    for f in _CNX_FIELDS:
        t = tensors.get(f)
        setattr(s, f, None if t is None else t.data_ptr())
    return s


def ln16_jvp(n, rstd, xdot, out=None):
    """Tangent of the first LayerNorm: ``n``, ``rstd`` from ln16 (or the fused epilogue), ``xdot`` the raw tangent."""
    assert n.is_contiguous() and xdot.is_contiguous() and xdot.shape == n.shape and xdot.dtype == n.dtype
    assert rstd.dtype == torch.float32 and rstd.is_contiguous() and rstd.numel() * 16 == n.numel()
    if out is None:
        out = torch.empty_like(xdot)
    _lib.check(_lib.lib().mfc_ln16_jvp(_lib.dtype_code(n.dtype), n.numel() // 16, n.data_ptr(), rstd.data_ptr(),
                                       xdot.data_ptr(), out.data_ptr(), _lib.stream_ptr()), "mfc_ln16_jvp")
    return out


def cnx_check_weights(w: dict, dtype) -> None:
    shapes = {"conv_w": (3, 3, 16, 16), "conv_b": (16,), "exp_w": (16, 32), "exp_b": (32,),
              "grn_gamma": (32,), "grn_beta": (32,), "con_w": (32, 16), "con_b": (16,), "ls": (16,)}
    for k, shp in shapes.items():
        t = w[k]
        want = dtype if k.endswith("_w") else torch.float32
        if tuple(t.shape)[-len(shp):] != shp and t.numel() != int(torch.tensor(shp).prod()):
            raise _lib.MfcError(f"ConvNeXt weight {k}: shape {tuple(t.shape)} != {shp} "
                                "(only C = 16 channels is implemented: condition_dimension >= 64)")
        if t.dtype != want or not t.is_contiguous() or not t.is_cuda:
            raise _lib.MfcError(f"ConvNeXt weight {k}: need contiguous device {want}, got {t.dtype}")


def _film(x, R):
    assert x.dtype == torch.float32 and x.shape == (R, 16) and x.is_contiguous()
    return x


def ln16(x, want_rstd=True):
    """Standalone first LayerNorm: (h1, rho0) from a raw [R, s, s, 16] map (mfc_ln16_fwd)."""
    assert x.is_contiguous() and x.numel() % 16 == 0
    npix = x.numel() // 16
    y = torch.empty_like(x)
    rstd = torch.empty(npix, dtype=torch.float32, device=x.device) if want_rstd else None
    _lib.check(_lib.lib().mfc_ln16_fwd(_lib.dtype_code(x.dtype), npix, x.data_ptr(), y.data_ptr(), _lib.ptr(rstd),
                                       _lib.stream_ptr()), "mfc_ln16_fwd")
    return y, rstd


_cnx_ws_cache: dict = {}


def cnx_workspace(R: int, s: int, device) -> torch.Tensor:
    """The partial-sum workspace of the ConvNeXt kernels (``mfc_cnx_ws_elems`` floats): one buffer per device, grown
    on demand, shared by every call -- launches on one stream are ordered, and each call consumes its records before
    it returns control to the next launch (the fixed-order reductions run inside the same C call)."""
    need = int(_lib.lib().mfc_cnx_ws_elems(R, s))
    if need <= 0:
        raise _lib.MfcError(f"mfc_cnx_ws_elems({R}, {s}) failed")
    key = (torch.device(device), torch.cuda.current_stream(device).cuda_stream)
    t = _cnx_ws_cache.get(key)
    if t is None or t.numel() < need:
        t = torch.empty(need, dtype=torch.float32, device=device)
        _cnx_ws_cache[key] = t
    return t


def cnx_forward(h0, scale, shift, w: dict, s: int, h0dot=None, scaledot=None, shiftdot=None, out=None,
                outdot=None, use_grn: bool = True, keep=None):
    """o = ConvNeXtBlock(FiLM(h1)) on [R, s, s, 16] where ``h0`` holds h1 = LN(h0) and ``h0dot`` the
    tangent of that LayerNorm (see ln16 / ln16_jvp, or the MFC_GEMM_LN16 / LN16T epilogue);
    returns (o, odot, G, q).  Runs mfc_cnx_stats -> mfc_grn_finalize -> mfc_cnx_apply.

    ``keep = (n1, rho1[, n1dot])`` (buffers [R, s, s, 16] in h0's dtype and fp32 [R, s, s]): the statistics pass writes
    n1 = LN(conv(FiLM(h1))), its 1/sigma and (tangent rows) the tangent of n1 there (``mfc_cnx_stats_save``); the apply
    pass then starts from them (``mfc_cnx_apply_n1``: no second conv / LayerNorm, bit-identical results), and
    ``cnx_backward(.., n1=, rho1=)`` can too."""
    _lib.require_cuda(h0)
    R = h0.shape[0]
    dt = _lib.dtype_code(h0.dtype)
    assert h0.is_contiguous() and h0.numel() == R * s * s * 16
    cnx_check_weights(w, h0.dtype)
    _film(scale, R), _film(shift, R)
    jvp = h0dot is not None
    if jvp:
        assert h0dot.is_contiguous() and h0dot.shape == h0.shape and h0dot.dtype == h0.dtype
        _film(scaledot, R), _film(shiftdot, R)
    L = _lib.lib()
    st = _lib.stream_ptr()
    ps = _cnx_struct(w)
    dev = h0.device
    S = torch.empty((2 if jvp else 1, R, 32), dtype=torch.float32, device=dev)
    ws = cnx_workspace(R, s, dev)
    G = torch.empty((R, 32), dtype=torch.float32, device=dev)
    q = torch.empty_like(G)
    qd = torch.empty_like(G) if jvp else None
    if keep is not None and not use_grn:
        raise _lib.MfcError("keep=(n1, rho1) needs the statistics pass (use_grn=True)")
    n1d = None
    if keep is not None:
        n1, rho1 = keep[0], keep[1]
        n1d = keep[2] if len(keep) > 2 else None         # scratch for the tangent of n1 (tangent rows)
        assert n1.shape == h0.shape and n1.dtype == h0.dtype and n1.is_contiguous()
        assert rho1.dtype == torch.float32 and rho1.is_contiguous() and rho1.numel() == R * s * s
        if jvp and n1d is None:
            n1d = torch.empty_like(h0)
        if jvp:
            assert n1d.shape == h0.shape and n1d.dtype == h0.dtype and n1d.is_contiguous()
        _lib.check(L.mfc_cnx_stats_save(dt, R, s, h0.data_ptr(), _lib.ptr(h0dot), scale.data_ptr(), shift.data_ptr(),
                                        _lib.ptr(scaledot), _lib.ptr(shiftdot), ctypes.byref(ps), S[0].data_ptr(),
                                        S[1].data_ptr() if jvp else None, ws.data_ptr(), n1.data_ptr(), rho1.data_ptr(),
                                        n1d.data_ptr() if jvp else None, st), "mfc_cnx_stats_save")
    elif use_grn:
        _lib.check(L.mfc_cnx_stats(dt, R, s, h0.data_ptr(), _lib.ptr(h0dot), scale.data_ptr(), shift.data_ptr(),
                                   _lib.ptr(scaledot), _lib.ptr(shiftdot), ctypes.byref(ps), S[0].data_ptr(),
                                   S[1].data_ptr() if jvp else None, ws.data_ptr(), st), "mfc_cnx_stats")
    if use_grn:
        _lib.check(L.mfc_grn_finalize(R, S[0].data_ptr(), S[1].data_ptr() if jvp else None, G.data_ptr(),
                                      q.data_ptr(), _lib.ptr(qd), st), "mfc_grn_finalize")
    else:
        # no GlobalResponseNormalization (conv_flow.py:91-92): the caller passes gamma = 1, beta = 0; with q = qdot = 0 the
        # apply pass computes y = g, and no statistics pass is needed
        G.zero_(); q.zero_()
        if jvp:
            qd.zero_()
    o = out if out is not None else torch.empty_like(h0)
    od = (outdot if outdot is not None else torch.empty_like(h0)) if jvp else None
    if keep is not None:
        _lib.check(L.mfc_cnx_apply_n1(dt, R, s, keep[0].data_ptr(), n1d.data_ptr() if jvp else None, h0.data_ptr(),
                                      _lib.ptr(h0dot), scale.data_ptr(), shift.data_ptr(), _lib.ptr(scaledot),
                                      _lib.ptr(shiftdot), ctypes.byref(ps), q.data_ptr(), _lib.ptr(qd), o.data_ptr(),
                                      _lib.ptr(od), st), "mfc_cnx_apply_n1")
        return o, od, G, q
    _lib.check(L.mfc_cnx_apply(dt, R, s, h0.data_ptr(), _lib.ptr(h0dot), scale.data_ptr(), shift.data_ptr(),
                               _lib.ptr(scaledot), _lib.ptr(shiftdot), ctypes.byref(ps), q.data_ptr(),
                               _lib.ptr(qd), o.data_ptr(), _lib.ptr(od), st), "mfc_cnx_apply")
    return o, od, G, q


def cnx_backward(h0, scale, shift, w: dict, s: int, G, q, dout, grads: dict, dh0=None, scratch=None, rho0=None,
                 use_grn: bool = True, n1=None, rho1=None):
    """Backward of cnx_forward's primal (``h0`` = h1 = LN(h0), ``rho0`` its 1/sigma): returns
    (dh0, dscale, dshift) with dh0 the gradient w.r.t. the RAW h0 (LayerNorm backward included);
    accumulates (+=) the small-parameter gradients into the fp32 tensors of ``grads``.
    ``n1``, ``rho1`` (what ``cnx_forward(.., keep=)`` kept): the first two passes start from them
    (``mfc_cnx_bwd_stats_n1`` / ``mfc_cnx_bwd_main_n1``) instead of repeating the conv and the LayerNorm."""
    assert rho0 is not None and rho0.dtype == torch.float32 and rho0.is_contiguous()
    _lib.require_cuda(h0, dout)
    R = h0.shape[0]
    dt = _lib.dtype_code(h0.dtype)
    assert dout.is_contiguous() and dout.shape == h0.shape and dout.dtype == h0.dtype
    for k in _CNX_FIELDS:
        g = grads[k]
        assert g.dtype == torch.float32 and g.is_contiguous() and g.numel() == w[k].numel(), k
    L = _lib.lib()
    st = _lib.stream_ptr()
    ps, gs = _cnx_struct(w), _cnx_struct(grads)
    dev = h0.device
    dq = torch.empty((R, 32), dtype=torch.float32, device=dev)
    kG = torch.empty_like(dq)
    ws = cnx_workspace(R, s, dev)
    from_n1 = n1 is not None
    if from_n1:
        assert rho1 is not None and n1.shape == h0.shape and n1.dtype == h0.dtype and n1.is_contiguous()
        assert rho1.dtype == torch.float32 and rho1.is_contiguous() and rho1.numel() == R * s * s
    if use_grn and from_n1:
        _lib.check(L.mfc_cnx_bwd_stats_n1(dt, R, s, n1.data_ptr(), ctypes.byref(ps), q.data_ptr(), dout.data_ptr(),
                                          dq.data_ptr(), ws.data_ptr(), st), "mfc_cnx_bwd_stats_n1")
    elif use_grn:
        _lib.check(L.mfc_cnx_bwd_stats(dt, R, s, h0.data_ptr(), scale.data_ptr(), shift.data_ptr(), ctypes.byref(ps),
                                       q.data_ptr(), dout.data_ptr(), dq.data_ptr(), ws.data_ptr(), st),
                   "mfc_cnx_bwd_stats")
    if use_grn:
        _lib.check(L.mfc_grn_bwd_finalize(R, G.data_ptr(), dq.data_ptr(), kG.data_ptr(),
                                          grads["grn_gamma"].data_ptr(), st), "mfc_grn_bwd_finalize")
    else:
        kG.zero_()          # no GRN: d g = d y (gamma + q) + g kG with gamma = 1, q = 0, kG = 0
    dc1 = scratch if scratch is not None else torch.empty_like(h0)
    if from_n1:
        _lib.check(L.mfc_cnx_bwd_main_n1(dt, R, s, n1.data_ptr(), rho1.data_ptr(), ctypes.byref(ps), q.data_ptr(),
                                         kG.data_ptr(), dout.data_ptr(), dc1.data_ptr(), ctypes.byref(gs), ws.data_ptr(), st),
                   "mfc_cnx_bwd_main_n1")
    else:
        _lib.check(L.mfc_cnx_bwd_main(dt, R, s, h0.data_ptr(), scale.data_ptr(), shift.data_ptr(), ctypes.byref(ps),
                                      q.data_ptr(), kG.data_ptr(), dout.data_ptr(), dc1.data_ptr(), ctypes.byref(gs),
                                      ws.data_ptr(), st), "mfc_cnx_bwd_main")
    if dh0 is None:
        dh0 = torch.empty_like(h0)
    dsc = torch.empty((R, 16), dtype=torch.float32, device=dev)
    dsh = torch.empty_like(dsc)
    _lib.check(L.mfc_cnx_bwd_conv(dt, R, s, h0.data_ptr(), rho0.data_ptr(), scale.data_ptr(), shift.data_ptr(), ctypes.byref(ps),
                                  dc1.data_ptr(), dout.data_ptr(), dh0.data_ptr(), ctypes.byref(gs),
                                  dsc.data_ptr(), dsh.data_ptr(), ws.data_ptr(), st), "mfc_cnx_bwd_conv")
    return dh0, dsc, dsh


# ---------------------------------------------------------------------------
# element-wise
# ---------------------------------------------------------------------------


def time_embed(t, h, dim, add=None, want_dot=False, tdot=None, hdot=None):
    _lib.require_cuda(t, h)
    R = t.numel()
    assert t.dtype == torch.float32 and h.dtype == torch.float32 and h.numel() == R
    cond = torch.empty((R, dim), dtype=torch.float32, device=t.device)
    cdot = torch.empty_like(cond) if want_dot else None
    if add is not None:
        assert add.shape == (R, dim) and add.dtype == torch.float32 and add.is_contiguous()
    _lib.check(_lib.lib().mfc_time_embed(R, dim, t.data_ptr(), h.data_ptr(), _lib.ptr(tdot), _lib.ptr(hdot),
                                         _lib.ptr(add), cond.data_ptr(), _lib.ptr(cdot), _lib.stream_ptr()),
               "mfc_time_embed")
    return cond, cdot


def gelu_fwd(pre, act_rows=None):
    M, N = pre.shape
    out = torch.empty_like(pre)
    _lib.check(_lib.lib().mfc_gelu_fwd(_lib.dtype_code(pre.dtype), M, N, M if act_rows is None else act_rows,
                                       pre.data_ptr(), out.data_ptr(), _lib.stream_ptr()), "mfc_gelu_fwd")
    return out


def gelu_bwd(pre, dout):
    assert pre.shape == dout.shape and pre.is_contiguous() and dout.is_contiguous()
    din = torch.empty_like(dout)
    _lib.check(_lib.lib().mfc_gelu_bwd(_lib.dtype_code(pre.dtype), pre.numel(), pre.data_ptr(), dout.data_ptr(),
                                       din.data_ptr(), _lib.stream_ptr()), "mfc_gelu_bwd")
    return din


def colsum(X, scale=1.0, out=None, accumulate=False):
    M, N = X.shape
    assert X.stride(1) == 1
    if out is None:
        out = torch.empty(N, dtype=torch.float32, device=X.device)
    dt = _lib.dtype_code(X.dtype)
    need = int(_lib.lib().mfc_colsum_ws_elems(dt, M, N)) if M >= 2048 else 0
    if need > 0 and (X.stride(0) * X.element_size()) % 16 == 0 and X.data_ptr() % 16 == 0:
        # tall activations (the Mixer's [B * tokens, C] maps): row-parallel two-stage sum, fixed order
        ws = torch.empty(need, dtype=torch.float32, device=X.device)
        _lib.check(_lib.lib().mfc_colsum_tall(dt, M, N, X.data_ptr(), X.stride(0), float(scale), out.data_ptr(),
                                              int(accumulate), ws.data_ptr(), _lib.stream_ptr()), "mfc_colsum_tall")
        return out
    _lib.check(_lib.lib().mfc_colsum(_lib.dtype_code(X.dtype), M, N, X.data_ptr(), X.stride(0), float(scale),
                                     out.data_ptr(), int(accumulate), _lib.stream_ptr()), "mfc_colsum")
    return out


def axpby(a, x, b=0.0, y=None, out=None):
    assert x.is_contiguous() and (y is None or (y.is_contiguous() and y.shape == x.shape and y.dtype == x.dtype))
    if out is None:
        out = torch.empty_like(x)
    _lib.check(_lib.lib().mfc_axpby(_lib.dtype_code(x.dtype), x.numel(), float(a), x.data_ptr(), float(b),
                                    _lib.ptr(y), out.data_ptr(), _lib.stream_ptr()), "mfc_axpby")
    return out


def cast(x, dtype, out=None):
    assert x.is_contiguous()
    if out is None:
        out = torch.empty(x.shape, dtype=dtype, device=x.device)
    _lib.check(_lib.lib().mfc_cast(_lib.dtype_code(x.dtype), _lib.dtype_code(dtype), x.numel(), x.data_ptr(),
                                   out.data_ptr(), _lib.stream_ptr()), "mfc_cast")
    return out


def randn(seed, stream_id, row0, B, D, device="cuda"):
    out = torch.empty((B, D), dtype=torch.float32, device=device)
    _lib.check(_lib.lib().mfc_randn(seed, stream_id, row0, B, D, out.data_ptr(), _lib.stream_ptr()), "mfc_randn")
    return out


def randn_dev(seed, stream_id, row0_dev, advance, out):
    """``mfc_randn_dev``: N(0,1) into ``out`` [B, D] (fp32 or bf16) keyed by the global row counter ``row0_dev`` (an
    int64 device scalar), which is advanced by ``advance`` afterwards -- a graph-capturable noise draw."""
    B, D = out.shape
    assert row0_dev.dtype == torch.int64 and row0_dev.numel() == 1 and row0_dev.is_cuda and out.is_contiguous()
    _lib.check(_lib.lib().mfc_randn_dev(_lib.dtype_code(out.dtype), seed, stream_id, row0_dev.data_ptr(), int(advance), B, D,
                                        out.data_ptr(), _lib.stream_ptr()), "mfc_randn_dev")
    return out


def data_size_of(Bglobal: int, data_proportion: float) -> int:
    """``data_size = int(batch_size * data_proportion)`` of utils.sample_tr (utils.py:41), in Python double arithmetic
    like the reference -- the ONE place this integer is computed (kernel and host bookkeeping both take it from here)."""
    return int(Bglobal * data_proportion)


def sample_tr(seed, step, row0, B, Bglobal, mean, std, data_proportion, pair=True, device="cuda", row_stride=1):
    if row0 < 0 or row_stride < 1 or row0 + (B - 1) * row_stride >= Bglobal:
        raise ValueError(f"shard rows {row0} + i*{row_stride} (i < {B}) exceed the global batch {Bglobal}")
    t = torch.empty((B, 1), dtype=torch.float32, device=device)
    r = torch.empty((B, 1), dtype=torch.float32, device=device) if pair else None
    _lib.check(_lib.lib().mfc_sample_tr(seed, step, row0, row_stride, B, data_size_of(Bglobal, data_proportion),
                                        float(mean), float(std), int(pair), t.data_ptr(), _lib.ptr(r),
                                        _lib.stream_ptr()), "mfc_sample_tr")
    return t, r


def flow_prepare(x, t, dtype, noise_min, noise_max, e=None, seed=0, step=0, row0=0, want_e=False, row_stride=1,
                 out=None):
    """z [dtype], target [fp32] (and e if drawn here and want_e).  Local row b is global row row0 + b*row_stride of
    the Philox noise stream.  ``out=(z, target)``: write into these (contiguous row slices are fine)."""
    B, D = x.shape
    assert x.dtype == torch.float32 and x.is_contiguous() and t.numel() == B and t.dtype == torch.float32
    assert t.is_contiguous()
    if out is None:
        z = torch.empty((B, D), dtype=dtype, device=x.device)
        target = torch.empty((B, D), dtype=torch.float32, device=x.device)
    else:
        z, target = out
        assert z.shape == (B, D) and z.dtype == dtype and z.is_contiguous()
        assert target.shape == (B, D) and target.dtype == torch.float32 and target.is_contiguous()
    e_out = torch.empty_like(target) if (want_e and e is None) else None
    if e is not None:
        assert e.shape == x.shape and e.dtype == torch.float32 and e.is_contiguous()
    _lib.check(_lib.lib().mfc_flow_prepare(_lib.dtype_code(dtype), B, D, x.data_ptr(), _lib.ptr(e), t.data_ptr(),
                                           float(noise_min), float(noise_max), seed, step, row0, row_stride,
                                           z.data_ptr(), target.data_ptr(), _lib.ptr(e_out), _lib.stream_ptr()),
               "mfc_flow_prepare")
    return z, target, (e if e is not None else e_out)


def flow_loss(u, target, *, dudt=None, n_tan=0, t=None, r=None, kind=0, mode=0, p=1.0, c=1e-3, Bglobal=None,
              want_grad=True):
    """Returns (loss scalar tensor, du or None, per-example pe)."""
    B, D = u.shape
    assert u.is_contiguous() and target.shape == u.shape and target.dtype == torch.float32
    if dudt is not None:      # n_tan < 0: the LAST -n_tan rows carry the tangents (dudt row b - (B + n_tan))
        assert dudt.dtype == u.dtype and dudt.is_contiguous() and dudt.shape[0] >= abs(n_tan) and dudt.shape[1] == D
    dev = u.device
    pe = torch.empty(B, dtype=torch.float32, device=dev)
    seed = torch.empty(B, dtype=torch.float32, device=dev)
    loss = torch.empty((), dtype=torch.float32, device=dev)
    du = torch.empty_like(u) if want_grad else None
    ws = torch.empty(B * 256, dtype=torch.float32, device=dev)      # MFC_FLOW_LOSS_WS_PER_ROW partial sums per example
    _lib.check(_lib.lib().mfc_flow_loss(_lib.dtype_code(u.dtype), kind, mode, B, B if Bglobal is None else Bglobal,
                                        D, u.data_ptr(), _lib.ptr(dudt), n_tan, _lib.ptr(t), _lib.ptr(r),
                                        target.data_ptr(), float(p), float(c), pe.data_ptr(), seed.data_ptr(),
                                        loss.data_ptr(), _lib.ptr(du), ws.data_ptr(), _lib.stream_ptr()),
               "mfc_flow_loss")
    return loss, du, pe


def adamw(p, g, m, v, *, lr, wd, step, b1=0.9, b2=0.999, eps=1e-8, p_bf16=None, grad_scale=1.0):
    assert p.dtype == torch.float32 and m.dtype == torch.float32 and v.dtype == torch.float32
    assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
    assert g.numel() == p.numel() == m.numel() == v.numel()
    if p_bf16 is not None:
        assert p_bf16.dtype == torch.bfloat16 and p_bf16.is_contiguous() and p_bf16.numel() == p.numel()
    _lib.check(_lib.lib().mfc_adamw(_lib.dtype_code(g.dtype), p.numel(), p.data_ptr(), _lib.ptr(p_bf16),
                                    g.data_ptr(), float(grad_scale), m.data_ptr(), v.data_ptr(), float(lr),
                                    float(b1), float(b2), float(eps), float(wd), int(step), _lib.stream_ptr()),
               "mfc_adamw")


def adamw_multi_items(leaves):
    """Descriptor table for ``adamw_multi``: ``leaves`` = iterable of (p, g, m, v, p_bf16 or None).  The table holds raw
    device pointers: build it once for buffers that live as long as the train state (masters, moments, working copies
    and the persistent gradient buffers are updated in place) and keep the tensors alive beside it."""
    leaves = list(leaves)
    arr = (_lib.AdamwItem * max(1, len(leaves)))()
    for it, (p, g, m, v, w) in zip(arr, leaves):
        assert p.dtype == torch.float32 and m.dtype == torch.float32 and v.dtype == torch.float32
        assert p.is_contiguous() and g.is_contiguous() and m.is_contiguous() and v.is_contiguous()
        assert g.numel() == p.numel() == m.numel() == v.numel() and p.is_cuda
        if w is not None:
            assert w.dtype == torch.bfloat16 and w.is_contiguous() and w.numel() == p.numel()
        it.p, it.p_bf16, it.g, it.m, it.v = p.data_ptr(), _lib.ptr(w), g.data_ptr(), m.data_ptr(), v.data_ptr()
        it.n, it.grad_dtype = p.numel(), _lib.dtype_code(g.dtype)
    return arr, len(leaves)


def adamw_multi(items, n_items, *, lr, wd, step, b1=0.9, b2=0.999, eps=1e-8, grad_scale=1.0):
    """``mfc_adamw_multi``: the AdamW update of many small leaves in one launch per 48 leaves (same element arithmetic
    as ``adamw``)."""
    if n_items == 0:
        return
    _lib.check(_lib.lib().mfc_adamw_multi(int(n_items), ctypes.addressof(items), float(grad_scale), float(lr), float(b1),
                                          float(b2), float(eps), float(wd), int(step), _lib.stream_ptr()),
               "mfc_adamw_multi")


# ---------------------------------------------------------------------------
# AdaLN / gating / strided copies (2-D views: stride(1) == 1, stride(0) = leading dimension)
# ---------------------------------------------------------------------------


def _v2(t):
    assert t.dim() == 2 and t.stride(1) == 1, (t.shape, t.stride())
    return t


def adaln_fwd(x, scale, shift, act_rows=None, mod_div=1, out=None):
    _v2(x), _v2(scale), _v2(shift)
    rows, W = x.shape
    assert scale.stride(0) == shift.stride(0) and scale.dtype == x.dtype == shift.dtype
    if out is None:
        out = torch.empty((rows, W), dtype=x.dtype, device=x.device)
    _lib.check(_lib.lib().mfc_adaln_fwd(_lib.dtype_code(x.dtype), rows, rows if act_rows is None else act_rows, W,
                                        x.data_ptr(), x.stride(0), scale.data_ptr(), shift.data_ptr(),
                                        scale.stride(0), mod_div, out.data_ptr(), out.stride(0), _lib.stream_ptr()),
               "mfc_adaln_fwd")
    return out


def adaln_bwd(x, scale, dy, dscale, dshift, mod_div=1, dx=None):
    _v2(x), _v2(scale), _v2(dy), _v2(dscale), _v2(dshift)
    rows, W = x.shape
    assert dscale.stride(0) == dshift.stride(0)
    if dx is None:
        dx = torch.empty((rows, W), dtype=x.dtype, device=x.device)
    assert dx.stride(0) == x.stride(0) or True
    # the kernel writes dx with x's leading dimension: use a dense temp unless they agree
    tmp = dx if dx.stride(0) == x.stride(0) else torch.empty_strided((rows, W), (x.stride(0), 1), dtype=x.dtype,
                                                                      device=x.device)
    _lib.check(_lib.lib().mfc_adaln_bwd(_lib.dtype_code(x.dtype), rows, W, x.data_ptr(), x.stride(0),
                                        scale.data_ptr(), scale.stride(0), mod_div, dy.data_ptr(), dy.stride(0),
                                        tmp.data_ptr(), dscale.data_ptr(), dshift.data_ptr(), dscale.stride(0),
                                        _lib.stream_ptr()), "mfc_adaln_bwd")
    if tmp is not dx:
        copy2d(tmp, dx)
    return dx


def gate_fwd(o, s2, res, inv_k, out, act_rows=None):
    _v2(o), _v2(s2), _v2(res), _v2(out)
    rows, W = o.shape
    _lib.check(_lib.lib().mfc_gate_fwd(_lib.dtype_code(o.dtype), rows, rows if act_rows is None else act_rows, W,
                                       o.data_ptr(), o.stride(0), s2.data_ptr(), s2.stride(0), res.data_ptr(),
                                       res.stride(0), float(inv_k), out.data_ptr(), out.stride(0),
                                       _lib.stream_ptr()), "mfc_gate_fwd")
    return out


def gate_bwd(dy, o, s2, inv_k, ds2, do=None):
    _v2(dy), _v2(o), _v2(s2), _v2(ds2)
    rows, W = o.shape
    if do is None:
        do = torch.empty_strided((rows, W), (o.stride(0), 1), dtype=o.dtype, device=o.device)
    assert do.stride(0) == o.stride(0)
    _lib.check(_lib.lib().mfc_gate_bwd(_lib.dtype_code(o.dtype), rows, W, dy.data_ptr(), dy.stride(0), o.data_ptr(),
                                       o.stride(0), s2.data_ptr(), s2.stride(0), float(inv_k), do.data_ptr(),
                                       ds2.data_ptr(), ds2.stride(0), _lib.stream_ptr()), "mfc_gate_bwd")
    return do


def chanmlp_ok(C: int, H: int) -> bool:
    """Shapes the fused channel MLP takes (forward and reverse): 16 channels and a hidden width the reverse kernel
    accepts -- the library decides (``mfc_chanmlp_ws_elems`` returns MFC_ENOSYS otherwise: H in {128, 256, 512} or a
    multiple of 1024)."""
    return C == 16 and H > 0 and int(_lib.lib().mfc_chanmlp_ws_elems(16, int(H))) > 0


def chanmlp_fwd(a, W1, b1, W2, b2, act_rows=None, residual=None, out=None):
    """out = gelu(a W1 + b1) W2 + b2 + residual on [rows, 16] tokens (rows >= act_rows: tangents), the hidden
    activation kept in registers (mfc_chanmlp_fwd; models/mlp_mixer.py:66-94)."""
    rows, C = a.shape
    H = W1.shape[1]
    assert C == 16 and W1.shape == (16, H) and W2.shape == (H, 16) and a.is_contiguous()
    assert W1.dtype == a.dtype and W2.dtype == a.dtype and W1.is_contiguous() and W2.is_contiguous()
    assert b1.dtype == torch.float32 and b2.dtype == torch.float32
    if out is None:
        out = torch.empty_like(a)
    assert residual is None or (residual.shape == a.shape and residual.is_contiguous() and residual.dtype == a.dtype)
    _lib.check(_lib.lib().mfc_chanmlp_fwd(_lib.dtype_code(a.dtype), rows, rows if act_rows is None else act_rows, H,
                                          a.data_ptr(), W1.data_ptr(), b1.data_ptr(), W2.data_ptr(), b2.data_ptr(),
                                          _lib.ptr(residual), out.data_ptr(), _lib.stream_ptr()), "mfc_chanmlp_fwd")
    return out


def chanmlp_bwd(a, dy, W1, b1, W2, dW1, db1, dW2, da=None):
    """Reverse of chanmlp_fwd on the primal rows: returns da; dW1 / db1 / dW2 are overwritten (mfc_chanmlp_bwd)."""
    rows, C = a.shape
    H = W1.shape[1]
    assert C == 16 and dy.shape == a.shape and a.is_contiguous() and dy.is_contiguous() and dy.dtype == a.dtype
    assert dW1.shape == W1.shape and dW2.shape == W2.shape and dW1.dtype == a.dtype and dW2.dtype == a.dtype
    assert dW1.is_contiguous() and dW2.is_contiguous() and db1.dtype == torch.float32 and db1.numel() == H
    if da is None:
        da = torch.empty_like(a)
    n = _lib.lib().mfc_chanmlp_ws_elems(rows, H)
    if n < 0:
        raise _lib.MfcError(f"mfc_chanmlp_bwd: unsupported hidden width {H}")
    ws = torch.empty(n, dtype=torch.float32, device=a.device)      # caching allocator: stream-ordered reuse
    _lib.check(_lib.lib().mfc_chanmlp_bwd(_lib.dtype_code(a.dtype), rows, H, a.data_ptr(), dy.data_ptr(), W1.data_ptr(),
                                          b1.data_ptr(), W2.data_ptr(), da.data_ptr(), dW1.data_ptr(), db1.data_ptr(),
                                          dW2.data_ptr(), ws.data_ptr(), _lib.stream_ptr()), "mfc_chanmlp_bwd")
    return da


def copy2d(src, dst, alpha=1.0, accumulate=False):
    _v2(src), _v2(dst)
    assert src.shape == dst.shape and src.dtype == dst.dtype
    rows, W = src.shape
    _lib.check(_lib.lib().mfc_copy2d(_lib.dtype_code(src.dtype), rows, W, src.data_ptr(), src.stride(0),
                                     dst.data_ptr(), dst.stride(0), float(alpha), int(accumulate),
                                     _lib.stream_ptr()), "mfc_copy2d")
    return dst


def transpose(src, batch, rows, cols, add=None, alpha=1.0, out=None):
    """[batch, rows, cols] -> [batch, cols, rows] (+ add), contiguous buffers of batch*rows*cols elements."""
    assert src.is_contiguous() and src.numel() == batch * rows * cols
    if out is None:
        out = torch.empty(src.numel(), dtype=src.dtype, device=src.device)
    assert out.is_contiguous() and out.numel() == src.numel() and out.dtype == src.dtype
    if add is not None:
        assert add.is_contiguous() and add.numel() == src.numel() and add.dtype == src.dtype
    _lib.check(_lib.lib().mfc_transpose(_lib.dtype_code(src.dtype), batch, rows, cols, src.data_ptr(), out.data_ptr(),
                                        float(alpha), _lib.ptr(add), _lib.stream_ptr()), "mfc_transpose")
    return out
