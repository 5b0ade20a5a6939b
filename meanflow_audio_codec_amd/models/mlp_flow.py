"""ConditionalFlow (MLP / AdaLN) on MI355X -- host mirror of ``models/mlp_flow.py``.

``MLP`` (:12-31), ``MLPEncoder`` (:39-55), ``ConditionalResidualBlock`` (:63-117) and ``ConditionalFlow``
(:125-230) with the reference's constructor arguments, parameter names/layouts and
``apply`` signature; the arithmetic runs in ``mfc_gemm``, ``mfc_adaln_*``, ``mfc_gate_*``, ``mfc_gelu_*``.
Like the ConvNeXt flow it exposes explicit primal / primal+tangent / reverse passes for the loss
strategies.  ``[latent, x]`` concatenation (:190-194) is never materialised separately: every block
writes its output straight into the noise columns of the next block's ``[R, L+D]`` input.
"""
from __future__ import annotations

import torch

from .. import _lib, ops
from .common import dense, dense_dw, dense_dx, init_from_shapes


class MlpCtx:
    __slots__ = ("R", "blocks", "cstack", "enc")

    def __init__(self):
        self.R, self.blocks, self.cstack, self.enc = 0, [], None, None


class ConditionalFlow:
    def __init__(self, noise_dimension: int, condition_dimension: int, num_blocks: int, latent_dimension: int,
                 dtype: torch.dtype = torch.float32):
        if condition_dimension % 2:
            raise ValueError(f"condition_dimension must be even, got {condition_dimension}")
        self.noise_dimension = noise_dimension
        self.condition_dimension = condition_dimension
        self.num_blocks = num_blocks
        self.latent_dimension = latent_dimension
        self.input_dimension = latent_dimension + noise_dimension
        self.dtype = dtype

    # ------------------------------------------------------------------ params
    def param_shapes(self) -> dict:
        D, L, Cd, I = self.noise_dimension, self.latent_dimension, self.condition_dimension, self.input_dimension
        H = (D + L) // 2
        sh = {"encoder/encoder_mlp/dense1/kernel": (D, H), "encoder/encoder_mlp/dense1/bias": (H,),
              "encoder/encoder_mlp/dense2/kernel": (H, L), "encoder/encoder_mlp/dense2/bias": (L,)}
        for i in range(self.num_blocks):
            b = f"blocks_{i}"
            sh[f"{b}/conditioning_layer/dense1/kernel"] = (Cd, Cd); sh[f"{b}/conditioning_layer/dense1/bias"] = (Cd,)
            sh[f"{b}/conditioning_layer/dense2/kernel"] = (Cd, 2 * I + D)
            sh[f"{b}/conditioning_layer/dense2/bias"] = (2 * I + D,)
            sh[f"{b}/mlp/dense1/kernel"] = (I, I); sh[f"{b}/mlp/dense1/bias"] = (I,)
            sh[f"{b}/mlp/dense2/kernel"] = (I, D); sh[f"{b}/mlp/dense2/bias"] = (D,)
        return sh

    def init(self, seed: int = 0, device="cuda") -> dict:
        return init_from_shapes(self.param_shapes(), seed, device)

    def compute_dtype_of(self, name: str) -> torch.dtype:
        return self.dtype if name.endswith("/kernel") else torch.float32

    def new_ctx(self) -> MlpCtx:
        return MlpCtx()

    def release_workspace(self):
        pass

    def _T(self, t):
        t = t.contiguous()
        return t if t.dtype == self.dtype else ops.cast(t, self.dtype)

    # ------------------------------------------------------------------ conditioning
    def encode(self, w: dict, x: torch.Tensor, ctx: MlpCtx | None = None) -> torch.Tensor:
        """ConditionalFlow.encode -> MLPEncoder (models/mlp_flow.py:39-55,153-162)."""
        xt = self._T(x)
        p = "encoder/encoder_mlp"
        a = dense(xt, w[f"{p}/dense1/kernel"], w[f"{p}/dense1/bias"])
        g = ops.gelu_fwd(a)
        lat = dense(g, w[f"{p}/dense2/kernel"], w[f"{p}/dense2/bias"])
        if ctx is not None:
            ctx.enc = (xt, a, g)
        return lat

    @property
    def latent_shape(self) -> tuple:
        """Per-sample shape of the latents ``encode`` returns (what ``sample`` callers must feed)."""
        return (self.latent_dimension,)

    def conditioning(self, w: dict, t, h, latents, want_dot: bool = False):
        """cond = emb(t) + emb(h) (models/mlp_flow.py:181-183); latents enter by concatenation, not here."""
        return ops.time_embed(t.reshape(-1).contiguous(), h.reshape(-1).contiguous(), self.condition_dimension,
                              want_dot=want_dot)

    # ------------------------------------------------------------------ passes
    def forward(self, w: dict, x, cond, *, xdot=None, cond_dot=None, latents=None, save: bool = False,
                ctx: MlpCtx | None = None):
        _lib.require_cuda(x, cond)
        R, D = x.shape
        L, I, K, T, dev = self.latent_dimension, self.input_dimension, self.num_blocks, self.dtype, x.device
        n_tan = 0 if xdot is None else xdot.shape[0]
        Rt = R + n_tan
        cs = cond if n_tan == 0 else torch.cat([cond, cond_dot[:n_tan]], 0)
        cstack = self._T(cs)
        XC = torch.zeros((Rt, I), dtype=T, device=dev)          # zero latents when latents is None (:223-228)
        if latents is not None:
            ops.copy2d(self._T(latents.reshape(R, -1)), XC[:R, :L])
        ops.copy2d(x, XC[:R, L:])
        if n_tan:
            ops.copy2d(xdot, XC[R:, L:])
        if save:
            ctx = ctx or MlpCtx()
            ctx.R, ctx.cstack, ctx.blocks = R, cstack, []
        for i in range(K):
            b = f"blocks_{i}"
            ac = dense(cstack, w[f"{b}/conditioning_layer/dense1/kernel"], w[f"{b}/conditioning_layer/dense1/bias"],
                       bias_rows=R)
            c1 = ops.gelu_fwd(ac, act_rows=R)
            sss = dense(c1, w[f"{b}/conditioning_layer/dense2/kernel"], w[f"{b}/conditioning_layer/dense2/bias"],
                        bias_rows=R)
            s1, sh, s2 = sss[:, :I], sss[:, I:2 * I], sss[:, 2 * I:]
            xn = ops.adaln_fwd(XC, s1, sh, act_rows=R)
            hh = dense(xn, w[f"{b}/mlp/dense1/kernel"], w[f"{b}/mlp/dense1/bias"], bias_rows=R)
            g = ops.gelu_fwd(hh, act_rows=R)
            o = dense(g, w[f"{b}/mlp/dense2/kernel"], w[f"{b}/mlp/dense2/bias"], bias_rows=R)
            XCn = torch.empty((Rt, I), dtype=T, device=dev)
            ops.copy2d(XC[:, :L], XCn[:, :L])
            ops.gate_fwd(o, s2, XC[:, L:], 1.0 / K, XCn[:, L:], act_rows=R)
            if save:
                ctx.blocks.append((XC, ac, c1, sss, xn, hh, g, o))
            XC = XCn
        out = torch.empty((R, D), dtype=T, device=dev)
        ops.copy2d(XC[:R, L:], out)
        outdot = None
        if n_tan:
            outdot = torch.empty((n_tan, D), dtype=T, device=dev)
            ops.copy2d(XC[R:, L:], outdot)
        return out, outdot, (ctx if save else None)

    def backward(self, w: dict, ctx: MlpCtx, dout, grads: dict, on_block=None, fused=None):
        R, K, T = ctx.R, self.num_blocks, self.dtype
        D, L, I, dev = self.noise_dimension, self.latent_dimension, self.input_dimension, dout.device
        dXC = torch.zeros((R, I), dtype=T, device=dev)
        ops.copy2d(dout, dXC[:, L:])
        dlat = torch.zeros((R, L), dtype=T, device=dev)
        dcond = torch.zeros((R, self.condition_dimension), dtype=T, device=dev)
        cst = ctx.cstack[:R]
        for i in reversed(range(K)):
            b = f"blocks_{i}"
            XC, ac, c1, sss, xn, hh, g, o = ctx.blocks[i]
            XC, ac, c1, sss, xn, hh, g, o = XC[:R], ac[:R], c1[:R], sss[:R], xn[:R], hh[:R], g[:R], o[:R]
            ops.copy2d(dXC[:, :L], dlat, accumulate=True)        # block i+1's input carried a copy of the latents
            dy = dXC[:, L:]
            dsss = torch.empty((R, 2 * I + D), dtype=T, device=dev)
            do = ops.gate_bwd(dy, o, sss[:, 2 * I:], 1.0 / K, dsss[:, 2 * I:])
            dense_dw(g, do, out=grads[f"{b}/mlp/dense2/kernel"])
            ops.colsum(do, out=grads[f"{b}/mlp/dense2/bias"])
            dg = dense_dx(do, w[f"{b}/mlp/dense2/kernel"])
            dh = ops.gelu_bwd(hh.contiguous(), dg)
            dense_dw(xn, dh, out=grads[f"{b}/mlp/dense1/kernel"])
            ops.colsum(dh, out=grads[f"{b}/mlp/dense1/bias"])
            dxn = dense_dx(dh, w[f"{b}/mlp/dense1/kernel"])
            dXCi = torch.empty((R, I), dtype=T, device=dev)
            ops.adaln_bwd(XC, sss[:, :I], dxn, dsss[:, :I], dsss[:, I:2 * I], dx=dXCi)
            ops.copy2d(dy, dXCi[:, L:], accumulate=True)          # residual: out = ... + x[:, -D:]
            # conditioning MLP
            dense_dw(c1, dsss, out=grads[f"{b}/conditioning_layer/dense2/kernel"])
            ops.colsum(dsss, out=grads[f"{b}/conditioning_layer/dense2/bias"])
            dc1 = dense_dx(dsss, w[f"{b}/conditioning_layer/dense2/kernel"])
            dac = ops.gelu_bwd(ac.contiguous(), dc1)
            dense_dw(cst, dac, out=grads[f"{b}/conditioning_layer/dense1/kernel"])
            ops.colsum(dac, out=grads[f"{b}/conditioning_layer/dense1/bias"])
            dcond = dense_dx(dac, w[f"{b}/conditioning_layer/dense1/kernel"], residual=dcond, beta=1.0)
            dXC = dXCi
            if on_block is not None:
                on_block([k for k in grads if k.startswith(b + "/")])
        ops.copy2d(dXC[:, :L], dlat, accumulate=True)
        dx = torch.empty((R, D), dtype=T, device=dev)
        ops.copy2d(dXC[:, L:], dx)
        dcond32 = dcond if dcond.dtype == torch.float32 else ops.cast(dcond, torch.float32)
        return dx, dcond32, dlat

    def backward_conditioning(self, w: dict, ctx: MlpCtx, dcond, latents, grads: dict, dlat=None):
        """Encoder gradients from d(latents) (the conditioning vector itself has no parameters here)."""
        if ctx.enc is None or dlat is None:
            return
        xt, a, g = ctx.enc
        p = "encoder/encoder_mlp"
        dense_dw(g, dlat, out=grads[f"{p}/dense2/kernel"])
        ops.colsum(dlat, out=grads[f"{p}/dense2/bias"])
        dg = dense_dx(dlat, w[f"{p}/dense2/kernel"])
        da = ops.gelu_bwd(a, dg)
        dense_dw(xt, da, out=grads[f"{p}/dense1/kernel"])
        ops.colsum(da, out=grads[f"{p}/dense1/bias"])

    def apply(self, variables: dict, x, time=None, latents=None, method: str | None = None):
        w = variables["params"]
        if method == "encode":
            return self.encode(w, x)
        cond, _ = self.conditioning(w, time[:, 0].contiguous(), time[:, 1].contiguous(), latents)
        out, _, _ = self.forward(w, self._T(x), cond, latents=latents)
        return out

    __call__ = apply
