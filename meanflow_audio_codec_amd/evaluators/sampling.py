"""Sampler -- host mirror of ``evaluators/sampling.py:5-97`` plus the true 1-NFE MeanFlow decode.

``sample(apply_fn, noise_dimension, params, key, latents, n_steps, use_improved_mean_flow,
guidance_scale)`` keeps the reference signature: Heun integration t: 1 -> 0 with h = 0 (NFE =
2 n_steps, x2 with CFG).  ``one_step_decode`` is the 1-NFE path the reference only documents
(documentation/research/improved_meanflow/improved_meanflow_key_eqn.md:313-316):
x0 = eps - u(eps, r=0, t=1), i.e. model time input [t=1, h=1].

Both are static-shape launch sequences, so ``GraphedDecoder`` captures them (and the IMDCT) into a
hipGraph and replays it (``torch.cuda.CUDAGraph`` is only the capture/replay plumbing; every node is
one of this library's kernels).
"""
from __future__ import annotations

import torch

from .. import ops
from ..trainers.time_sampling import PRNGKey


def _model_of(apply_fn):
    m = getattr(apply_fn, "__self__", None)
    if m is None:
        raise TypeError("apply_fn must be the bound `apply` of a meanflow_audio_codec_amd model")
    return m


def _velocity(model, w, x, tval: float, latents, guidance_scale: float):
    B = x.shape[0]
    t = torch.full((B,), float(tval), dtype=torch.float32, device=x.device)
    h = torch.zeros_like(t)
    cond, _ = model.conditioning(w, t, h, latents)
    k, _, _ = model.forward(w, x, cond, latents=latents)
    if guidance_scale != 1.0:
        k = k.clone()
        cond0, _ = model.conditioning(w, t, h, None)
        k0, _, _ = model.forward(w, x, cond0, latents=None)
        k = ops.axpby(guidance_scale, k, 1.0 - guidance_scale, k0)
    return k


def heun_integrate(model, w, x, latents, n_steps: int, guidance_scale: float = 1.0):
    """evaluators/sampling.py:52-96 from a given start x (model dtype)."""
    dt = 1.0 / float(n_steps)
    ts = torch.linspace(1.0, 0.0, n_steps, dtype=torch.float32).tolist()
    for t in ts:
        k1 = _velocity(model, w, x, t, latents, guidance_scale).clone()
        x2 = ops.axpby(1.0, x, -dt, k1)
        k2 = _velocity(model, w, x2, t - dt, latents, guidance_scale)
        ksum = ops.axpby(1.0, k1, 1.0, k2)
        x = ops.axpby(1.0, x, -dt / 2.0, ksum)
    return x


def sample(apply_fn, noise_dimension: int, params: dict, key, latents: torch.Tensor | None = None,
           n_steps: int = 100, use_improved_mean_flow: bool = False, guidance_scale: float = 1.0) -> torch.Tensor:
    if latents is None:
        if guidance_scale != 1.0:
            raise ValueError("guidance_scale != 1.0 requires latents to be provided")
        raise ValueError("latents must be provided for conditional sampling")
    model = _model_of(apply_fn)
    if not isinstance(key, PRNGKey):
        key = PRNGKey(int(key))
    B = latents.shape[0]
    x = ops.randn(key.seed, 0x5a00 + (key.counter & 0xFF), 0, B, noise_dimension, device=latents.device)
    x = x if model.dtype == torch.float32 else ops.cast(x, model.dtype)
    out = heun_integrate(model, params, x, latents, n_steps, guidance_scale)
    return out if out.dtype == torch.float32 else ops.cast(out.contiguous(), torch.float32)


def one_step_decode(model, w, eps: torch.Tensor, latents: torch.Tensor | None) -> torch.Tensor:
    """x0 = eps - u(eps, [t=1, h=1], latents); eps in the model dtype, result too."""
    B = eps.shape[0]
    one = torch.ones((B,), dtype=torch.float32, device=eps.device)
    cond, _ = model.conditioning(w, one, one, latents)
    u, _, _ = model.forward(w, eps, cond, latents=latents)
    return ops.axpby(1.0, eps, -1.0, u)


class GraphedDecoder:
    """hipGraph of: Philox noise -> n-step Heun (or the 1-NFE decode) -> un-flatten -> IMDCT.

    ``n_steps == 0`` selects the 1-NFE decode.  Static shapes: batch B, latents fixed at capture.  The noise draw is
    a node of the graph (``mfc_randn_dev``: the global row counter lives in a device scalar that the graph advances by
    B after every draw), so ONE replay is the whole "noise -> audio" path and consecutive replays decode fresh noise;
    ``fresh_noise=False`` rewinds the counter first and so repeats the previous draw."""

    noise_in_graph = True

    def __init__(self, model, w, B: int, latents: torch.Tensor | None, *, n_steps: int = 0,
                 token_shape: tuple[int, int] | None = None, mdct_config=None, seed: int = 0, device="cuda"):
        from ..preprocessing.mdct import imdct
        self.model, self.w, self.B, self.n_steps = model, w, B, n_steps
        self.latents = latents
        self.eps = torch.empty((B, model.noise_dimension), dtype=model.dtype, device=device)   # the start noise of the last replay
        self.row0 = torch.zeros(1, dtype=torch.int64, device=device)     # first global row of the NEXT draw
        self.seed = seed
        self.calls = 0
        self._imdct = imdct

        def body():
            e = ops.randn_dev(self.seed, 0x6400, self.row0, B, self.eps)
            if n_steps == 0:
                x0 = one_step_decode(model, w, e, latents)
            else:
                x0 = heun_integrate(model, w, e, latents, n_steps)
            x0 = x0 if x0.dtype == torch.float32 else ops.cast(x0.contiguous(), torch.float32)
            if token_shape is not None:
                return self._imdct(x0.reshape(B, token_shape[0], token_shape[1]), config=mdct_config)
            return x0

        self._body = body
        getattr(w, "wait_all", lambda: None)()   # no in-flight weight gathers may be awaited inside the capture
        body()                                   # warm-up: allocations, function attributes
        torch.cuda.synchronize()
        self.row0.zero_()                        # the warm-up's draw does not count
        self.graph = torch.cuda.CUDAGraph()
        # The cyclic garbage collector must not run inside the capture: collecting an unrelated object that owns device
        # resources (an older hipGraph, an event) issues HIP calls that are illegal while a stream is capturing and
        # aborts the process (seen as "Fatal Python error: Aborted ... Garbage-collecting" in a test run).
        import gc
        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            with torch.cuda.graph(self.graph):
                self.out = body()
        finally:
            if was_enabled:
                gc.enable()

    def noise_of_call(self, k: int) -> torch.Tensor:
        """The fp32 start noise replay number ``k`` (0-based) decodes: rows [k B, (k+1) B) of Philox stream 0x6400."""
        return ops.randn(self.seed, 0x6400, k * self.B, self.B, self.model.noise_dimension, device=self.eps.device)

    def __call__(self, fresh_noise: bool = True) -> torch.Tensor:
        if not fresh_noise and self.calls > 0:
            self.row0.sub_(self.B)               # repeat the previous draw
            self.calls -= 1
        self.graph.replay()
        self.calls += 1
        return self.out
