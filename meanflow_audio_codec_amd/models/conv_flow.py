"""ConditionalConvFlow on MI355X -- host mirror of ``models/conv_flow.py``.

Same constructor arguments and call signature as the reference
(``ConditionalConvFlow`` :213-271, ``ConditionalConvNeXtBlock`` :123-205,
``ConvNeXtBlock`` :53-115, ``GlobalResponseNormalization`` :14-45); all arithmetic
runs in the HIP kernels behind the C ABI (``mfc_gemm``, ``mfc_cnx_*``,
``mfc_time_embed``, ``mfc_gelu_*``).  Because there is no tracing autodiff here,
the model also exposes the three passes the loss strategies need:

* ``forward(..)``            primal only (v pass, sampling)
* ``forward(.., xdot=..)``   primal + forward-mode tangent, row-stacked ``[x; xdot]`` so
                             every weight tile is read once
* ``backward(ctx, dout)``    reverse pass through the saved primal

Deliberate fixes of reference defects: the model gets an ``encode``
method (defect 2; a bottleneck encoder, unpinned -- see ``encode``); ``latent_proj`` is
created at init time when ``latent_input_dim`` is given (defect 12).
"""
from __future__ import annotations

import math
import os

import torch

from .. import _lib, ops
from .common import dense, dense_dw, dense_dx, init_from_shapes

BOTTLENECK = 128  # models/conv_flow.py:142,153


class ConvCtx:
    """Saved primal activations of one forward pass (consumed by ``backward``)."""
    __slots__ = ("R", "x_in", "a1", "g1", "a2", "g2", "H0", "rho", "O", "G", "q", "sc", "sh", "cond", "lat", "enc", "N1",
                 "rho1", "stride")

    def __init__(self):
        for k in self.__slots__:
            setattr(self, k, None)


class ConditionalConvFlow:
    def __init__(self, noise_dimension: int, condition_dimension: int, num_blocks: int, latent_dimension: int,
                 image_size: int = 28, use_grn: bool = True, num_latent_tokens: int = 32,
                 latent_input_dim: int | None = None, dtype: torch.dtype = torch.float32):
        if condition_dimension % 2:
            raise ValueError(f"condition_dimension must be even, got {condition_dimension}")
        self.noise_dimension = noise_dimension
        self.condition_dimension = condition_dimension
        self.num_blocks = num_blocks
        self.latent_dimension = latent_dimension
        self.num_latent_tokens = num_latent_tokens
        self.use_grn = bool(use_grn)               # models/conv_flow.py:91-92: False skips GlobalResponseNormalization
        self.spatial_size = int(math.sqrt(noise_dimension))            # conv_flow.py:138
        self.channels = min(16, condition_dimension // 4)              # conv_flow.py:139
        if self.channels != 16:
            raise _lib.MfcError("HIP ConvNeXt kernels implement C = 16 channels (condition_dimension >= 64); "
                                f"got C = {self.channels}")
        self.S = self.spatial_size ** 2 * self.channels
        # the reference's train_flow feeds [B, latent_dimension] latents (trainers/train.py:367-370);
        # proj/mnist_trial feeds [B, 32, latent] (:163-166).  latent_proj is sized for what is fed.
        self.latent_input_dim = latent_dimension if latent_input_dim is None else latent_input_dim
        self.dtype = dtype
        # keep n1 = LN(conv(FiLM(h1))) of the statistics pass for the apply and reverse passes (streaming kernels instead
        # of a second / third / fourth conv + LayerNorm); costs one more [R, S] map per saved block
        self.keep_n1 = os.environ.get("MFC_CNX_KEEP_N1", "1") != "0"
        self._ws = {}
        self._block_names = [self.block_param_names(i) for i in range(num_blocks)]

    # ------------------------------------------------------------------ params
    def param_shapes(self) -> dict:
        D, S, Cd, C = self.noise_dimension, self.S, self.condition_dimension, self.channels
        sh = {}
        for i in range(self.num_blocks):
            b = f"blocks_{i}"
            sh[f"{b}/input_proj1/kernel"] = (D, BOTTLENECK); sh[f"{b}/input_proj1/bias"] = (BOTTLENECK,)
            sh[f"{b}/input_proj2/kernel"] = (BOTTLENECK, S); sh[f"{b}/input_proj2/bias"] = (S,)
            sh[f"{b}/conditioning_layer/kernel"] = (Cd, 2 * C); sh[f"{b}/conditioning_layer/bias"] = (2 * C,)
            cb = f"{b}/conv_block"
            sh[f"{cb}/Conv_0/kernel"] = (3, 3, C, C); sh[f"{cb}/Conv_0/bias"] = (C,)
            sh[f"{cb}/Conv_1/kernel"] = (1, 1, C, 2 * C); sh[f"{cb}/Conv_1/bias"] = (2 * C,)
            if self.use_grn:
                sh[f"{cb}/GlobalResponseNormalization_0/gamma"] = (2 * C,)
                sh[f"{cb}/GlobalResponseNormalization_0/beta"] = (2 * C,)
            sh[f"{cb}/Conv_2/kernel"] = (1, 1, 2 * C, C); sh[f"{cb}/Conv_2/bias"] = (C,)
            sh[f"{cb}/layer_scale_gamma"] = (C,)
            sh[f"{b}/output_proj1/kernel"] = (S, BOTTLENECK); sh[f"{b}/output_proj1/bias"] = (BOTTLENECK,)
            sh[f"{b}/output_proj2/kernel"] = (BOTTLENECK, D); sh[f"{b}/output_proj2/bias"] = (D,)
        sh["latent_proj/kernel"] = (self.latent_input_dim, Cd); sh["latent_proj/bias"] = (Cd,)
        sh["encoder/dense1/kernel"] = (D, BOTTLENECK); sh["encoder/dense1/bias"] = (BOTTLENECK,)
        sh["encoder/dense2/kernel"] = (BOTTLENECK, self.latent_dimension)
        sh["encoder/dense2/bias"] = (self.latent_dimension,)
        return sh

    def init(self, seed: int = 0, device="cuda") -> dict:
        return init_from_shapes(self.param_shapes(), seed, device)

    def compute_dtype_of(self, name: str) -> torch.dtype:
        """Kernels the big GEMMs / MFMA conv read live in the model dtype; the conditioning
        path (conditioning_layer, latent_proj, encoder/dense2) and every vector stay fp32."""
        if not name.endswith("/kernel"):
            return torch.float32
        small = ("conditioning_layer/kernel", "latent_proj/kernel", "encoder/dense2/kernel")
        return torch.float32 if name.endswith(small) else self.dtype

    # ------------------------------------------------------------------ helpers
    def _cnx_w(self, w: dict, i: int, grads: bool = False) -> dict:
        cb = f"blocks_{i}/conv_block"
        dev = w[f"{cb}/Conv_0/bias"].device
        if self.use_grn:
            gam, bet = w[f"{cb}/GlobalResponseNormalization_0/gamma"], w[f"{cb}/GlobalResponseNormalization_0/beta"]
        elif grads:
            # use_grn=False: the kernels still emit d gamma / d beta; they land in scratch nobody reads
            gam, bet = self._buf(("nogrn_dg",), (32,), torch.float32, dev), self._buf(("nogrn_db",), (32,), torch.float32, dev)
        else:
            # use_grn=False: y = g * (gamma + q) + beta with gamma = 1, beta = 0, q = 0 is the identity on g
            gam = self._ws.get(("nogrn_one", dev))
            if gam is None:
                gam = self._ws[("nogrn_one", dev)] = torch.ones(32, dtype=torch.float32, device=dev)
            bet = self._ws.get(("nogrn_zero", dev))
            if bet is None:
                bet = self._ws[("nogrn_zero", dev)] = torch.zeros(32, dtype=torch.float32, device=dev)
        return {"conv_w": w[f"{cb}/Conv_0/kernel"], "conv_b": w[f"{cb}/Conv_0/bias"],
                "exp_w": w[f"{cb}/Conv_1/kernel"], "exp_b": w[f"{cb}/Conv_1/bias"],
                "grn_gamma": gam, "grn_beta": bet,
                "con_w": w[f"{cb}/Conv_2/kernel"], "con_b": w[f"{cb}/Conv_2/bias"],
                "ls": w[f"{cb}/layer_scale_gamma"]}

    def _cnx_g(self, g: dict, i: int) -> dict:
        return self._cnx_w(g, i, grads=True)

    def _buf(self, key, shape, dtype, device):
        t = self._ws.get(key)
        if t is None or t.shape != tuple(shape) or t.dtype != dtype or t.device != device:
            t = torch.empty(shape, dtype=dtype, device=device)
            self._ws[key] = t
        return t

    def release_workspace(self):
        self._ws.clear()

    # ------------------------------------------------------------------ conditioning
    def encode(self, w: dict, x: torch.Tensor, ctx: ConvCtx | None = None) -> torch.Tensor:
        """BUILD DECISION (reference defect 2): ``ConditionalConvFlow`` has no ``encode`` although every
        loss strategy calls ``apply_fn(.., method="encode")`` (trainers/loss_strategies.py:99,167,250), and
        the reference's encoders (MLPEncoder D->(D+L)/2, MLPMixerEncoder D->512*L) are 10^10-parameter
        layers at D = 392704.  Here: Dense(D->128) -> GELU -> Dense(128->latent), the block's own
        bottleneck pattern (models/conv_flow.py:142-146).  Unpinned by the reference."""
        xt = x if x.dtype == self.dtype else ops.cast(x.contiguous(), self.dtype)
        a = dense(xt, w["encoder/dense1/kernel"], w["encoder/dense1/bias"])
        g = ops.gelu_fwd(a)
        g32 = g if g.dtype == torch.float32 else ops.cast(g, torch.float32)
        lat = dense(g32, w["encoder/dense2/kernel"], w["encoder/dense2/bias"])
        if ctx is not None:
            ctx.enc = (xt, a, g32)
        return lat

    @property
    def latent_shape(self) -> tuple:
        """Per-sample shape of the latents ``encode`` returns and ``latent_proj`` is sized for."""
        return (self.latent_input_dim,)

    def conditioning(self, w: dict, t: torch.Tensor, h: torch.Tensor, latents: torch.Tensor | None,
                     want_dot: bool = False):
        """cond = emb(t) + emb(h) (+ latent_proj(flatten(latents))), models/conv_flow.py:257-267; the
        tangent w.r.t. (t, h) with (tdot, hdot) = (1, 1) when ``want_dot`` (latents carry no tangent)."""
        add = None
        if latents is not None:
            lf = latents.reshape(latents.shape[0], -1).to(torch.float32).contiguous()
            if lf.shape[1] != self.latent_input_dim:
                raise ValueError(f"latents flatten to {lf.shape[1]} features, latent_proj expects "
                                 f"{self.latent_input_dim}")
            add = dense(lf, w["latent_proj/kernel"], w["latent_proj/bias"])
        return ops.time_embed(t.reshape(-1).contiguous(), h.reshape(-1).contiguous(),
                              self.condition_dimension, add=add, want_dot=want_dot)

    # ------------------------------------------------------------------ passes
    def new_ctx(self) -> ConvCtx:
        return ConvCtx()

    def forward(self, w: dict, x: torch.Tensor, cond: torch.Tensor, *, xdot: torch.Tensor | None = None,
                cond_dot: torch.Tensor | None = None, latents=None, save: bool = False, ctx: ConvCtx | None = None):
        """Velocity net on R rows.  ``xdot`` ([n_tan, D], n_tan <= R) carries the tangents of the FIRST
        n_tan rows; returns (out [R,D], outdot [n_tan,D] | None, ctx | None)."""
        _lib.require_cuda(x, cond)
        R, D = x.shape
        assert D == self.noise_dimension and x.dtype == self.dtype and cond.shape == (R, self.condition_dimension)
        n_tan = 0 if xdot is None else xdot.shape[0]
        assert n_tan <= R
        Rt = R + n_tan
        K, S, dev, T = self.num_blocks, self.S, x.device, self.dtype
        keep_n1 = self.use_grn and self.keep_n1
        X = self._buf(("X0", Rt), (Rt, D), T, dev) if not save else torch.empty((Rt, D), dtype=T, device=dev)
        X[:R].copy_(x)
        if n_tan:
            X[R:].copy_(xdot)
            cstack = torch.cat([cond, cond_dot[:n_tan]], 0).contiguous()
        else:
            cstack = cond
        if not save:
            ctx = None
        if save:
            ctx = ctx or ConvCtx()
            ctx.R = R
            ctx.x_in, ctx.a1, ctx.g1, ctx.a2, ctx.g2, ctx.G, ctx.q, ctx.sc, ctx.sh = ([] for _ in range(9))
            ctx.cond = cond
            # block i primal rows live at [i*R, (i+1)*R); its tangent rows spill into the head of block
            # i+1's region and are dead before that region is written.
            ctx.H0 = self._buf(("H0save", R, n_tan), (K * R + n_tan, S), T, dev)    # holds h1 = LN(h0) (primal rows)
            ctx.O = self._buf(("Osave", R, n_tan), (K * R + n_tan, S), T, dev)
            ctx.rho = self._buf(("rhosave", R), (K, R, S // 16), torch.float32, dev)
            ctx.stride = R
            if keep_n1:
                # n1 = LN(conv(FiLM(h1))) and its 1/sigma of every block's primal rows, written by the statistics pass:
                # the apply pass and the reverse pass start from them (ops.cnx_forward keep= / cnx_backward n1=)
                ctx.N1 = self._buf(("N1save", R), (K * R, S), T, dev)
                ctx.rho1 = self._buf(("rho1save", R), (K, R, S // 16), torch.float32, dev)

            def bufs(i):
                return (ctx.H0[i * R:i * R + Rt], ctx.O[i * R:i * R + Rt], ctx.rho[i],
                        ctx.N1[i * R:(i + 1) * R] if keep_n1 else None, ctx.rho1[i].view(R, -1) if keep_n1 else None)

            def record(i, X_in, a1, g1, a2, g2, G, q, sc, sh):
                ctx.x_in.append(X_in); ctx.a1.append(a1); ctx.g1.append(g1); ctx.a2.append(a2); ctx.g2.append(g2)
                ctx.G.append(G); ctx.q.append(q); ctx.sc.append(sc); ctx.sh.append(sh)
        else:
            H0s = self._buf(("H0", Rt), (Rt, S), T, dev)
            Os = self._buf(("O", Rt), (Rt, S), T, dev)
            rhos = self._buf(("rho", R), (R, S // 16), torch.float32, dev)
            N1s = self._buf(("N1", R), (R, S), T, dev) if keep_n1 else None
            rho1s = self._buf(("rho1", R), (R, S // 16), torch.float32, dev) if keep_n1 else None
            bufs = lambda i: (H0s, Os, rhos, N1s, rho1s)
            record = None
        X = self._run_blocks(w, X, cstack, R, n_tan, bufs, record, save)
        out = X[:R]
        outdot = X[R:] if n_tan else None
        return out, outdot, ctx

    def _run_blocks(self, w, X, cstack, R, n_tan, bufs, record, fresh_x):
        """The K blocks on the row-stacked [primal (R); tangent (n_tan)] batch ``X``.  ``bufs(i)`` -> (H0 [Rt,S],
        O [Rt,S], rho [R, s*s], N1 [R,S] | None, rho1 [R, s*s] | None) of block i; ``record(i, ...)`` keeps what the
        reverse pass needs; ``fresh_x``: every block output gets its own tensor (it is a saved block input)."""
        K, S, s, dev, T, D = self.num_blocks, self.S, self.spatial_size, X.device, self.dtype, self.noise_dimension
        Rt = R + n_tan
        for i in range(K):
            b = f"blocks_{i}"
            a1 = dense(X, w[f"{b}/input_proj1/kernel"], w[f"{b}/input_proj1/bias"], bias_rows=R)
            g1 = ops.gelu_fwd(a1, act_rows=R)
            H0, O, rho, N1, r1 = bufs(i)
            # h0 = g1 W2 + b2 with the block's first LayerNorm fused into the epilogue: the primal rows of H0
            # hold h1 = LN(h0), rho its per-pixel 1/sigma; the tangent rows get the tangent of that LayerNorm
            ops.gemm(g1, w[f"{b}/input_proj2/kernel"], bias=w[f"{b}/input_proj2/bias"], bias_rows=R, out=H0,
                     ln_rstd=rho, ln_tangent=True)
            cp = dense(cstack, w[f"{b}/conditioning_layer/kernel"], w[f"{b}/conditioning_layer/bias"], bias_rows=R)
            sc, sh = cp[:R, :16].contiguous(), cp[:R, 16:].contiguous()
            cw = self._cnx_w(w, i)
            Gs, qs = [], []
            if N1 is not None:
                N1d = self._buf(("N1dot", n_tan), (n_tan, S), T, dev) if n_tan else None     # tangent of n1: scratch
                keep = lambda a, b: (N1[a:b], r1[a:b]) if a else (N1[a:b], r1[a:b], N1d)
            else:
                keep = lambda a, b: None
            if n_tan:
                scd, shd = cp[R:, :16].contiguous(), cp[R:, 16:].contiguous()
                _, _, G, q = ops.cnx_forward(H0[:n_tan], sc[:n_tan], sh[:n_tan], cw, s, h0dot=H0[R:],
                                             scaledot=scd, shiftdot=shd, out=O[:n_tan], outdot=O[R:],
                                             use_grn=self.use_grn, keep=keep(0, n_tan))
                Gs.append(G); qs.append(q)
            if R > n_tan:
                _, _, G, q = ops.cnx_forward(H0[n_tan:R], sc[n_tan:], sh[n_tan:], cw, s, out=O[n_tan:R],
                                             use_grn=self.use_grn, keep=keep(n_tan, R))
                Gs.append(G); qs.append(q)
            a2 = dense(O, w[f"{b}/output_proj1/kernel"], w[f"{b}/output_proj1/bias"], bias_rows=R)
            g2 = ops.gelu_fwd(a2, act_rows=R)
            Xn = torch.empty((Rt, D), dtype=T, device=dev) if fresh_x else self._buf(("X", i & 1, Rt), (Rt, D), T, dev)
            dense(g2, w[f"{b}/output_proj2/kernel"], w[f"{b}/output_proj2/bias"], bias_rows=R, out=Xn,
                  alpha=1.0 / K, residual=X, beta=1.0)
            if record is not None:
                record(i, X, a1, g1, a2, g2, torch.cat(Gs, 0) if len(Gs) > 1 else Gs[0],
                       torch.cat(qs, 0) if len(qs) > 1 else qs[0], sc, sh)
            X = Xn
        return X

    def forward_imf(self, w: dict, z: torch.Tensor, cond_u: torch.Tensor, cond_v: torch.Tensor, cdot: torch.Tensor,
                    n_plain: int, ctx: ConvCtx | None = None):
        """The two forward passes of the improved-MeanFlow loss (trainers/loss_strategies.py:253-270) with the rows that
        need no tangent riding along with the boundary pass.  Rows of ``z`` / ``cond_u``: [plain (r == t): n_plain;
        tangent: n_tan]; ``cond_v`` / ``cdot`` ([n_tan, cond]) belong to the tangent rows.

          pass A (R rows, primal only):  [plain rows with cond_u ; tangent rows with cond_v]  ->  u_plain, v
          pass B (n_tan + n_tan rows):   [tangent rows with cond_u ; their tangents (v, cdot)]  ->  u_tan, du/dt

        instead of a 64-row boundary pass and a 192-row pass: every GEMM and ConvNeXt launch of the two passes works on
        128 rows (B = 128, p = 0.5).  Returns (u [R, D] in the row order of ``z``, dudt [n_tan, D], ctx).  The saved
        activations of all R rows end up in ONE context in that row order: a block's save region is
        [plain | tangent-primal | tangent-tangent] (stride R + n_tan rows), pass A writes [plain | v rows] into its
        first R rows, pass B then overwrites the v rows with its primal rows and appends its tangent rows."""
        _lib.require_cuda(z, cond_u, cond_v)
        R, D = z.shape
        n_tan = R - n_plain
        assert 0 < n_plain < R and cond_v.shape[0] == n_tan and cdot.shape[0] == n_tan and z.dtype == self.dtype
        K, S, dev, T = self.num_blocks, self.S, z.device, self.dtype
        keep_n1 = self.use_grn and self.keep_n1
        Rs = R + n_tan                                           # rows of one block's save region
        ctx = ctx or ConvCtx()
        ctx.R, ctx.stride, ctx.cond = R, Rs, cond_u
        ctx.H0 = self._buf(("H0save2", R, n_tan), (K * Rs, S), T, dev)
        ctx.O = self._buf(("Osave2", R, n_tan), (K * Rs, S), T, dev)
        ctx.rho = self._buf(("rhosave", R), (K, R, S // 16), torch.float32, dev)
        if keep_n1:
            ctx.N1 = self._buf(("N1save", R), (K * R, S), T, dev)
            ctx.rho1 = self._buf(("rho1save", R), (K, R, S // 16), torch.float32, dev)
        saved = []

        # ---- pass A
        def bufs_a(i):
            return (ctx.H0[i * Rs:i * Rs + R], ctx.O[i * Rs:i * Rs + R], ctx.rho[i],
                    ctx.N1[i * R:(i + 1) * R] if keep_n1 else None, ctx.rho1[i].view(R, -1) if keep_n1 else None)

        def record_a(i, X_in, a1, g1, a2, g2, G, q, sc, sh):
            saved.append([X_in, a1, g1, a2, g2, G, q, sc, sh])
        XA = torch.empty((R, D), dtype=T, device=dev)
        XA.copy_(z)
        cond_a = torch.cat([cond_u[:n_plain], cond_v], 0).contiguous()
        out_a = self._run_blocks(w, XA, cond_a, R, 0, bufs_a, record_a, True)
        v = out_a[n_plain:]

        # ---- pass B: its primal rows take the place of pass A's v rows in every saved tensor
        def bufs_b(i):
            lo = i * Rs + n_plain
            return (ctx.H0[lo:lo + 2 * n_tan], ctx.O[lo:lo + 2 * n_tan], ctx.rho[i][n_plain:],
                    ctx.N1[i * R + n_plain:(i + 1) * R] if keep_n1 else None,
                    ctx.rho1[i].view(R, -1)[n_plain:] if keep_n1 else None)

        def record_b(i, X_in, a1, g1, a2, g2, G, q, sc, sh):
            for dst, src in zip(saved[i], (X_in, a1, g1, a2, g2, G, q, sc, sh)):
                dst[n_plain:R].copy_(src[:n_tan])
        XB = torch.empty((2 * n_tan, D), dtype=T, device=dev)
        XB[:n_tan].copy_(z[n_plain:])
        XB[n_tan:].copy_(v)
        cond_b = torch.cat([cond_u[n_plain:], cdot], 0).contiguous()
        out_b = self._run_blocks(w, XB, cond_b, n_tan, n_tan, bufs_b, record_b, True)
        ctx.x_in, ctx.a1, ctx.g1, ctx.a2, ctx.g2, ctx.G, ctx.q, ctx.sc, ctx.sh = (list(c) for c in zip(*saved))
        u = torch.cat([out_a[:n_plain], out_b[:n_tan]], 0)
        return u, out_b[n_tan:], v, ctx

    def block_param_names(self, i: int) -> list:
        pre = f"blocks_{i}/"
        return [k for k in self.param_shapes() if k.startswith(pre)]

    def backward(self, w: dict, ctx: ConvCtx, dout: torch.Tensor, grads: dict, on_block=None, fused=None):
        """Reverse pass through the saved primal.  ``on_block(names)`` is called as soon as the gradients
        of one block are complete (lets the caller overlap the exchange / AdamW with the rest of the pass).  Writes every block parameter's gradient into
        ``grads`` (big kernels: overwritten in the model dtype; small fp32 leaves: overwritten too).
        ``fused`` (train_state.FusedUpdater, single-GPU steps): the four big kernels of a block are updated by the
        epilogue of their own weight-gradient product (``mfc_gemm_adamw``) and never appear in ``grads``; every
        product that READS such a kernel is therefore issued before the one that updates it.
        Returns (dx [R,D], dcond [R,cond] fp32)."""
        def dw(name, A, dY, alpha=1.0):
            """weight gradient of leaf ``name`` (= A^T dY) and the bias gradient of the same layer (column sums of dY):
            fused single-GPU steps get both from one kernel (mfc_gemm_adamw with its colsum output)."""
            bias = name[:-len("kernel")] + "bias"
            if fused is not None and fused.dw(name, A, dY, alpha, bias_out=grads[bias], bias_scale=alpha):
                return
            if fused is None or not fused.dw(name, A, dY, alpha):
                dense_dw(A, dY, alpha=alpha, out=grads[name])
            ops.colsum(dY, scale=alpha, out=grads[bias])
        R, K, S, s, T = ctx.R, self.num_blocks, self.S, self.spatial_size, self.dtype
        D, dev = self.noise_dimension, dout.device
        assert dout.shape == (R, D) and dout.dtype == T and dout.is_contiguous()
        dX = dout
        dcond = torch.zeros((R, self.condition_dimension), dtype=torch.float32, device=dev)
        dO = self._buf(("dO", R), (R, S), T, dev)
        dH0 = self._buf(("dH0", R), (R, S), T, dev)
        dC1 = self._buf(("dC1", R), (R, S), T, dev)
        for i in reversed(range(K)):
            b = f"blocks_{i}"
            x_in = ctx.x_in[i][:R]
            a1, g1, a2, g2 = ctx.a1[i][:R], ctx.g1[i][:R], ctx.a2[i][:R], ctx.g2[i][:R]
            st = ctx.stride or R                   # rows of one block's save region (forward: R; forward_imf: R + n_tan)
            H0, O = ctx.H0[i * st:i * st + R], ctx.O[i * st:i * st + R]
            # out = (g2 W4 + b4)/K + x
            dg2 = dense_dx(dX, w[f"{b}/output_proj2/kernel"], alpha=1.0 / K)
            dw(f"{b}/output_proj2/kernel", g2, dX, 1.0 / K)
            da2 = ops.gelu_bwd(a2, dg2)
            dense_dx(da2, w[f"{b}/output_proj1/kernel"], out=dO)
            dw(f"{b}/output_proj1/kernel", O, da2)
            # ConvNeXt interior
            cg = self._cnx_g(grads, i)
            for t_ in cg.values():
                t_.zero_()
            n1kw = {} if ctx.N1 is None else dict(n1=ctx.N1[i * R:(i + 1) * R], rho1=ctx.rho1[i])
            _, dsc, dsh = ops.cnx_backward(H0, ctx.sc[i], ctx.sh[i], self._cnx_w(w, i), s, ctx.G[i], ctx.q[i], dO, cg,
                                           dh0=dH0, scratch=dC1, rho0=ctx.rho[i], use_grn=self.use_grn, **n1kw)
            dcp = torch.cat([dsc, dsh], 1).contiguous()
            dense_dw(ctx.cond, dcp, out=grads[f"{b}/conditioning_layer/kernel"])
            ops.colsum(dcp, out=grads[f"{b}/conditioning_layer/bias"])
            dcond = dense_dx(dcp, w[f"{b}/conditioning_layer/kernel"], residual=dcond, beta=1.0)
            # h0 = g1 W2 + b2
            dg1 = dense_dx(dH0, w[f"{b}/input_proj2/kernel"])
            dw(f"{b}/input_proj2/kernel", g1, dH0)
            da1 = ops.gelu_bwd(a1, dg1)
            dX = dense_dx(da1, w[f"{b}/input_proj1/kernel"], residual=dX, beta=1.0)
            dw(f"{b}/input_proj1/kernel", x_in, da1)
            if on_block is not None:
                on_block(self._block_names[i])
        return dX, dcond, None

    def backward_conditioning(self, w: dict, ctx: ConvCtx, dcond: torch.Tensor, latents, grads: dict, dlat=None):
        """Gradients of latent_proj and of the bottleneck encoder from d(cond)."""
        lf = latents.reshape(latents.shape[0], -1).to(torch.float32).contiguous()
        dense_dw(lf, dcond, out=grads["latent_proj/kernel"])
        ops.colsum(dcond, out=grads["latent_proj/bias"])
        if ctx.enc is None:
            return
        xt, a, g32 = ctx.enc
        dlat = dense_dx(dcond, w["latent_proj/kernel"])
        dense_dw(g32, dlat, out=grads["encoder/dense2/kernel"])
        ops.colsum(dlat, out=grads["encoder/dense2/bias"])
        dg = dense_dx(dlat, w["encoder/dense2/kernel"])
        dgT = dg if a.dtype == torch.float32 else ops.cast(dg, a.dtype)
        da = ops.gelu_bwd(a, dgT)
        dense_dw(xt, da, out=grads["encoder/dense1/kernel"])
        ops.colsum(da, out=grads["encoder/dense1/bias"])

    # reference-style call: apply({"params": w}, x, time[B,2], latents) / method="encode"
    def apply(self, variables: dict, x: torch.Tensor, time: torch.Tensor | None = None,
              latents: torch.Tensor | None = None, method: str | None = None) -> torch.Tensor:
        w = variables["params"]
        if method == "encode":
            return self.encode(w, x)
        cond, _ = self.conditioning(w, time[:, 0].contiguous(), time[:, 1].contiguous(), latents)
        xt = x if x.dtype == self.dtype else ops.cast(x.contiguous(), self.dtype)
        out, _, _ = self.forward(w, xt.contiguous(), cond)
        return out.clone()

    __call__ = apply
