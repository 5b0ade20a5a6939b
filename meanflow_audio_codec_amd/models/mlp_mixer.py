"""ConditionalMLPMixerFlow + MLPMixerEncoder on MI355X -- host mirror of ``models/mlp_mixer.py``.

``MLPMixerBlock`` (:14-94), ``ConditionalMLPMixerBlock`` (:102-163), ``ConditionalMLPMixerFlow`` (:171-235)
and ``MLPMixerEncoder`` (:243-323) with the reference's constructor arguments and Flax parameter names
(``Dense_0..5`` in call order inside the ``@nn.compact`` block).  Arithmetic:
``mfc_gemm`` (token / channel MLPs, projections), ``mfc_adaln_*`` (LayerNorm over channels + AdaLN with
per-sample modulation broadcast over tokens), ``mfc_transpose`` (token <-> channel, residual add fused
into the way back), ``mfc_gelu_*``.  The encoder is the reference's own ``MLPMixerEncoder`` -- the only
component that produces the ``[B, num_latent_tokens, latent_dim]`` latents the Mixer / ConvNeXt flows'
docstrings expect; ``encode`` is wired to it here (reference defect 2).
"""
from __future__ import annotations

import math
import os

import torch

from .. import _lib, ops
from .common import dense, dense_dw, dense_dx, init_from_shapes


# MFC_MIXER_FUSED=0: the channel MLP as two GEMMs + gelu kernels (A/B runs and the cross-check test)
_FUSED_CHANNEL_MLP = os.environ.get("MFC_MIXER_FUSED", "1") != "0"


def _mixer_shapes(prefix, nt, C, cond_dim, tmd, cmd):
    return {f"{prefix}/Dense_0/kernel": (cond_dim, 2 * C), f"{prefix}/Dense_0/bias": (2 * C,),
            f"{prefix}/Dense_1/kernel": (nt, tmd), f"{prefix}/Dense_1/bias": (tmd,),
            f"{prefix}/Dense_2/kernel": (tmd, nt), f"{prefix}/Dense_2/bias": (nt,),
            f"{prefix}/Dense_3/kernel": (cond_dim, 2 * C), f"{prefix}/Dense_3/bias": (2 * C,),
            f"{prefix}/Dense_4/kernel": (C, cmd), f"{prefix}/Dense_4/bias": (cmd,),
            f"{prefix}/Dense_5/kernel": (cmd, C), f"{prefix}/Dense_5/bias": (C,)}


class _MixerBlock:
    """MLPMixerBlock on N samples (the first R primal, the rest tangents) of nt tokens x C channels,
    stored [N*nt, C] (token-major, channel fastest)."""

    def __init__(self, prefix: str, nt: int, C: int, dtype):
        self.p, self.nt, self.C, self.T = prefix, nt, C, dtype

    def _mod(self, w, name, cstack, R):
        m = dense(cstack, w[f"{self.p}/{name}/kernel"], w[f"{self.p}/{name}/bias"], bias_rows=R)   # fp32 [N, 2C]
        return m if self.T == torch.float32 else ops.cast(m, self.T)

    def forward(self, w, X, cstack, R, save):
        p, nt, C = self.p, self.nt, self.C
        N = X.shape[0] // nt
        mod1 = self._mod(w, "Dense_0", cstack, R)
        a = ops.adaln_fwd(X, mod1[:, :C], mod1[:, C:], act_rows=R * nt, mod_div=nt)
        aT = ops.transpose(a, N, nt, C).view(N * C, nt)
        h = dense(aT, w[f"{p}/Dense_1/kernel"], w[f"{p}/Dense_1/bias"], bias_rows=R * C)
        g = ops.gelu_fwd(h, act_rows=R * C)
        tk = dense(g, w[f"{p}/Dense_2/kernel"], w[f"{p}/Dense_2/bias"], bias_rows=R * C)       # [N*C, nt]
        x1 = ops.transpose(tk, N, C, nt, add=X).view(N * nt, C)                                 # + residual
        mod2 = self._mod(w, "Dense_3", cstack, R)
        a2 = ops.adaln_fwd(x1, mod2[:, :C], mod2[:, C:], act_rows=R * nt, mod_div=nt)
        W4, W5 = w[f"{p}/Dense_4/kernel"], w[f"{p}/Dense_5/kernel"]
        if self.fused_channel_mlp(W4):
            # 16-channel tokens: both Dense layers, the gelu and the residual in one kernel; the [N*nt, channel_mix_dim]
            # hidden activation is neither written nor saved (the reverse kernel recomputes it from a2)
            x2 = ops.chanmlp_fwd(a2, W4, w[f"{p}/Dense_4/bias"], W5, w[f"{p}/Dense_5/bias"], act_rows=R * nt, residual=x1)
            h2 = g2 = None
        else:
            h2 = dense(a2, W4, w[f"{p}/Dense_4/bias"], bias_rows=R * nt)
            g2 = ops.gelu_fwd(h2, act_rows=R * nt)
            x2 = dense(g2, W5, w[f"{p}/Dense_5/bias"], bias_rows=R * nt, residual=x1, beta=1.0)
        saved = (X, mod1, aT, h, g, x1, mod2, a2, h2, g2) if save else None
        return x2, saved

    def fused_channel_mlp(self, W4) -> bool:
        return _FUSED_CHANNEL_MLP and ops.chanmlp_ok(self.C, W4.shape[1]) and W4.dtype == self.T

    def backward(self, w, saved, dx2, cond, R, grads):
        """dx2 [R*nt, C] -> (dX [R*nt, C], dcond [R, cond] fp32); parameter gradients into ``grads``."""
        p, nt, C, T = self.p, self.nt, self.C, self.T
        X, mod1, aT, h, g, x1, mod2, a2, h2, g2 = saved
        X, x1, a2 = X[:R * nt], x1[:R * nt], a2[:R * nt]
        aT, h, g = aT[:R * C], h[:R * C], g[:R * C]
        dev = dx2.device
        # channel mixing: x2 = mlp(adaln(x1)) + x1
        if h2 is None:
            ops.colsum(dx2, out=grads[f"{p}/Dense_5/bias"])
            da2 = ops.chanmlp_bwd(a2, dx2.contiguous(), w[f"{p}/Dense_4/kernel"], w[f"{p}/Dense_4/bias"], w[f"{p}/Dense_5/kernel"],
                                  grads[f"{p}/Dense_4/kernel"], grads[f"{p}/Dense_4/bias"], grads[f"{p}/Dense_5/kernel"])
        else:
            h2, g2 = h2[:R * nt], g2[:R * nt]
            dense_dw(g2, dx2, out=grads[f"{p}/Dense_5/kernel"]); ops.colsum(dx2, out=grads[f"{p}/Dense_5/bias"])
            dh2 = ops.gelu_bwd(h2, dense_dx(dx2, w[f"{p}/Dense_5/kernel"]))
            dense_dw(a2, dh2, out=grads[f"{p}/Dense_4/kernel"]); ops.colsum(dh2, out=grads[f"{p}/Dense_4/bias"])
            da2 = dense_dx(dh2, w[f"{p}/Dense_4/kernel"])
        dmod2 = torch.empty((R, 2 * C), dtype=torch.float32, device=dev)
        dx1 = ops.adaln_bwd(x1, mod2[:R, :C], da2, dmod2[:, :C], dmod2[:, C:], mod_div=nt)
        dx1 = ops.axpby(1.0, dx1, 1.0, dx2)
        dense_dw(cond, dmod2, out=grads[f"{p}/Dense_3/kernel"]); ops.colsum(dmod2, out=grads[f"{p}/Dense_3/bias"])
        dcond = dense_dx(dmod2, w[f"{p}/Dense_3/kernel"])
        # token mixing: x1 = T(mlp(T(adaln(X)))) + X
        dtk = ops.transpose(dx1, R, nt, C).view(R * C, nt)
        dense_dw(g, dtk, out=grads[f"{p}/Dense_2/kernel"]); ops.colsum(dtk, out=grads[f"{p}/Dense_2/bias"])
        dh = ops.gelu_bwd(h, dense_dx(dtk, w[f"{p}/Dense_2/kernel"]))
        dense_dw(aT, dh, out=grads[f"{p}/Dense_1/kernel"]); ops.colsum(dh, out=grads[f"{p}/Dense_1/bias"])
        da = ops.transpose(dense_dx(dh, w[f"{p}/Dense_1/kernel"]), R, C, nt).view(R * nt, C)
        dmod1 = torch.empty((R, 2 * C), dtype=torch.float32, device=dev)
        dX = ops.adaln_bwd(X, mod1[:R, :C], da, dmod1[:, :C], dmod1[:, C:], mod_div=nt)
        dX = ops.axpby(1.0, dX, 1.0, dx1)
        dense_dw(cond, dmod1, out=grads[f"{p}/Dense_0/kernel"]); ops.colsum(dmod1, out=grads[f"{p}/Dense_0/bias"])
        dcond = dense_dx(dmod1, w[f"{p}/Dense_0/kernel"], residual=dcond, beta=1.0)
        return dX, dcond


class MixerCtx:
    __slots__ = ("R", "cond", "blocks", "enc")

    def __init__(self):
        self.R, self.cond, self.blocks, self.enc = 0, None, [], None


class ConditionalMLPMixerFlow:
    def __init__(self, noise_dimension: int, condition_dimension: int, num_blocks: int, latent_dimension: int,
                 token_mix_dim: int = 2048, channel_mix_dim: int = 2048, num_channels: int = 16,
                 num_latent_tokens: int = 32, num_context_tokens: int = 512, dtype: torch.dtype = torch.float32):
        if condition_dimension % 2:
            raise ValueError(f"condition_dimension must be even, got {condition_dimension}")
        self.noise_dimension, self.condition_dimension = noise_dimension, condition_dimension
        self.num_blocks, self.latent_dimension = num_blocks, latent_dimension
        self.token_mix_dim, self.channel_mix_dim, self.num_channels = token_mix_dim, channel_mix_dim, num_channels
        self.num_latent_tokens, self.num_context_tokens = num_latent_tokens, num_context_tokens
        self.spatial_size = int(math.sqrt(noise_dimension))          # mlp_mixer.py:118
        self.num_tokens = self.spatial_size ** 2
        self.dtype = dtype
        self.mix = [_MixerBlock(f"blocks_{i}/mixer_block", self.num_tokens, num_channels, dtype)
                    for i in range(num_blocks)]
        self.enc_mix = _MixerBlock("encoder/mixer_block", num_context_tokens + num_latent_tokens, latent_dimension, dtype)

    def param_shapes(self) -> dict:
        D, Cd, L, nt, C = (self.noise_dimension, self.condition_dimension, self.latent_dimension, self.num_tokens,
                           self.num_channels)
        sh = {}
        for i in range(self.num_blocks):
            b = f"blocks_{i}"
            sh[f"{b}/input_proj/kernel"] = (D, nt * C); sh[f"{b}/input_proj/bias"] = (nt * C,)
            sh.update(_mixer_shapes(f"{b}/mixer_block", nt, C, Cd, self.token_mix_dim, self.channel_mix_dim))
            sh[f"{b}/output_proj/kernel"] = (nt * C, D); sh[f"{b}/output_proj/bias"] = (D,)
        sh["latent_proj/kernel"] = (self.num_latent_tokens * L, Cd); sh["latent_proj/bias"] = (Cd,)
        sh["encoder/input_proj/kernel"] = (D, self.num_context_tokens * L)
        sh["encoder/input_proj/bias"] = (self.num_context_tokens * L,)
        sh["encoder/latent_queries"] = (self.num_latent_tokens, L)
        sh["encoder/condition_emb"] = (L,)
        sh.update(_mixer_shapes("encoder/mixer_block", self.num_context_tokens + self.num_latent_tokens, L, L,
                                self.token_mix_dim, self.channel_mix_dim))
        return sh

    def init(self, seed: int = 0, device="cuda") -> dict:
        p = init_from_shapes(self.param_shapes(), seed, device)
        gen = torch.Generator(device=device).manual_seed(seed + 1)
        for k in ("encoder/latent_queries", "encoder/condition_emb"):       # normal(0.02), mlp_mixer.py:261-273
            p[k].normal_(0.0, 0.02, generator=gen)
        return p

    def compute_dtype_of(self, name: str) -> torch.dtype:
        if not name.endswith("/kernel"):
            return torch.float32
        small = ("Dense_0/kernel", "Dense_3/kernel", "latent_proj/kernel")   # conditioning path stays fp32
        return torch.float32 if name.endswith(small) else self.dtype

    def new_ctx(self) -> MixerCtx:
        return MixerCtx()

    def release_workspace(self):
        pass

    def _T(self, t):
        t = t.contiguous()
        return t if t.dtype == self.dtype else ops.cast(t, self.dtype)

    # ------------------------------------------------------------------ conditioning
    def encode(self, w: dict, x, ctx: MixerCtx | None = None):
        """MLPMixerEncoder.__call__ (models/mlp_mixer.py:281-323) -> [B, num_latent_tokens, latent_dim]."""
        B, L = x.shape[0], self.latent_dimension
        n_ctx, n_lat = self.num_context_tokens, self.num_latent_tokens
        nt = n_ctx + n_lat
        xt = self._T(x)
        allt = torch.empty((B, nt * L), dtype=self.dtype, device=x.device)
        dense(xt, w["encoder/input_proj/kernel"], w["encoder/input_proj/bias"], out=allt[:, :n_ctx * L])
        q = self._T(w["encoder/latent_queries"].reshape(1, n_lat * L)).expand(B, n_lat * L)
        allt[:, n_ctx * L:].copy_(q)
        cond = w["encoder/condition_emb"].reshape(1, L).expand(B, L).contiguous()
        out, saved = self.enc_mix.forward(w, allt.view(B * nt, L), cond, B, ctx is not None)
        lat = out.view(B, nt, L)[:, n_ctx:, :].contiguous()
        if ctx is not None:
            ctx.enc = (xt, cond, saved)
        return lat

    @property
    def latent_shape(self) -> tuple:
        """Per-sample shape of the latents ``encode`` returns and ``latent_proj`` is sized for."""
        return (self.num_latent_tokens, self.latent_dimension)

    def conditioning(self, w: dict, t, h, latents, want_dot: bool = False):
        add = None
        if latents is not None:
            lf = latents.reshape(latents.shape[0], -1).to(torch.float32).contiguous()
            add = dense(lf, w["latent_proj/kernel"], w["latent_proj/bias"])
        return ops.time_embed(t.reshape(-1).contiguous(), h.reshape(-1).contiguous(), self.condition_dimension,
                              add=add, want_dot=want_dot)

    # ------------------------------------------------------------------ passes
    def forward(self, w: dict, x, cond, *, xdot=None, cond_dot=None, latents=None, save: bool = False,
                ctx: MixerCtx | None = None):
        _lib.require_cuda(x, cond)
        R, D = x.shape
        K, nt, C = self.num_blocks, self.num_tokens, self.num_channels
        n_tan = 0 if xdot is None else xdot.shape[0]
        N = R + n_tan
        cstack = cond if n_tan == 0 else torch.cat([cond, cond_dot[:n_tan]], 0).contiguous()
        X = torch.cat([x, xdot], 0) if n_tan else x
        if save:
            ctx = ctx or MixerCtx()
            ctx.R, ctx.cond, ctx.blocks = R, cond, []
        for i in range(K):
            b = f"blocks_{i}"
            P = dense(X, w[f"{b}/input_proj/kernel"], w[f"{b}/input_proj/bias"], bias_rows=R)
            m, saved = self.mix[i].forward(w, P.view(N * nt, C), cstack, R, save)
            Xn = dense(m.view(N, nt * C), w[f"{b}/output_proj/kernel"], w[f"{b}/output_proj/bias"], bias_rows=R,
                       alpha=1.0 / K, residual=X, beta=1.0)
            if save:
                ctx.blocks.append((X, m, saved))
            X = Xn
        return X[:R], (X[R:] if n_tan else None), (ctx if save else None)

    def backward(self, w: dict, ctx: MixerCtx, dout, grads: dict, on_block=None, fused=None):
        R, K, nt, C = ctx.R, self.num_blocks, self.num_tokens, self.num_channels
        dX = dout
        dcond = torch.zeros((R, self.condition_dimension), dtype=torch.float32, device=dout.device)
        for i in reversed(range(K)):
            b = f"blocks_{i}"
            X, m, saved = ctx.blocks[i]
            X, m = X[:R], m.view(-1, nt * C)[:R]
            dm = dense_dx(dX, w[f"{b}/output_proj/kernel"], alpha=1.0 / K)
            dense_dw(m, dX, alpha=1.0 / K, out=grads[f"{b}/output_proj/kernel"])
            ops.colsum(dX, scale=1.0 / K, out=grads[f"{b}/output_proj/bias"])
            dP, dc = self.mix[i].backward(w, saved, dm.view(R * nt, C), ctx.cond, R, grads)
            dcond = ops.axpby(1.0, dcond, 1.0, dc)
            dP = dP.view(R, nt * C)
            dense_dw(X, dP, out=grads[f"{b}/input_proj/kernel"])
            ops.colsum(dP, out=grads[f"{b}/input_proj/bias"])
            dX = dense_dx(dP, w[f"{b}/input_proj/kernel"], residual=dX, beta=1.0)
            if on_block is not None:
                on_block([k for k in grads if k.startswith(b + "/")])
        return dX, dcond, None

    def backward_conditioning(self, w: dict, ctx: MixerCtx, dcond, latents, grads: dict, dlat=None):
        lf = latents.reshape(latents.shape[0], -1).to(torch.float32).contiguous()
        dense_dw(lf, dcond, out=grads["latent_proj/kernel"])
        ops.colsum(dcond, out=grads["latent_proj/bias"])
        if ctx.enc is None:
            return
        xt, econd, saved = ctx.enc
        B, L = xt.shape[0], self.latent_dimension
        n_ctx, n_lat = self.num_context_tokens, self.num_latent_tokens
        nt = n_ctx + n_lat
        dl = dense_dx(dcond, w["latent_proj/kernel"])                      # fp32 [B, n_lat*L]
        dall = torch.zeros((B, nt * L), dtype=self.dtype, device=xt.device)
        ops.copy2d(self._T(dl), dall[:, n_ctx * L:])
        dA, dce = self.enc_mix.backward(w, saved, dall.view(B * nt, L), econd, B, grads)
        dA = dA.view(B, nt * L)
        ops.colsum(dce, out=grads["encoder/condition_emb"])
        ops.colsum(dA[:, n_ctx * L:], out=grads["encoder/latent_queries"].view(-1))
        dctx = dA[:, :n_ctx * L]
        dense_dw(xt, dctx, out=grads["encoder/input_proj/kernel"])
        ops.colsum(dctx, out=grads["encoder/input_proj/bias"])

    def apply(self, variables: dict, x, time=None, latents=None, method: str | None = None):
        w = variables["params"]
        if method == "encode":
            return self.encode(w, x)
        cond, _ = self.conditioning(w, time[:, 0].contiguous(), time[:, 1].contiguous(), latents)
        out, _, _ = self.forward(w, self._T(x), cond)
        return out.contiguous()

    __call__ = apply
