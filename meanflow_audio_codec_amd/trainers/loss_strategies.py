"""Loss strategies -- host mirror of ``trainers/loss_strategies.py``.

``LossStrategy.compute_loss(state, key, x) -> (loss, grads)`` with the reference's names and
constructor arguments.  There is no tracing autodiff: each strategy drives the model's explicit
passes (primal / primal+tangent / reverse), all of which run in HIP kernels:

* ImprovedMeanFlowLoss (:204-280): v = f(z,[t,0]) (no grad, only rows with r != t), then ONE
  row-stacked pass [z; zdot=v] for (u, dudt) with tangent (v, 1, 0) -> th tangent (1, 1),
  v_pred = u + (t-r) sg(dudt), weighted L2 / MSE, reverse pass through u only.
* MeanFlowLoss (:115-201): tangent (e-x, 1, 0), u_tgt = v - clip(t-r,0,1) sg(dudt), adaptive weight.
* FlowMatchingLoss (:50-112).

Rows whose r == t multiply dudt by exactly 0 (:270), so their v / tangent passes are skipped
(algorithmic 4 F_fwd instead of 5); rows are re-ordered so tangent rows come first.

Extra keyword arguments (not in the reference): ``e, t, r`` to pass the random draws explicitly
(parity tests), ``row0`` / ``row_stride`` / ``global_batch`` for data-parallel shards: local row i is
global row ``row0 + i * row_stride`` (means are over the GLOBAL batch and the deterministic "first
int(B*p) rows have r = t" rule of utils.sample_tr is applied per global row).  Interleaved ownership
(rank k owns rows k, k+G, ...: ``row0 = k, row_stride = G``) gives every rank the same number of
r == t rows (+-1), i.e. the same work; contiguous ownership (``row0 = k*B, row_stride = 1``) would give
the first half of the ranks only r == t rows (3 forward-equivalents) and the rest only tangent rows (5).
"""
from __future__ import annotations

import os
from abc import ABC, abstractmethod

import torch

from .. import ops
from .noise_schedules import LinearNoiseSchedule, NoiseSchedule
from .time_sampling import LogitNormalTimeSampling, MeanFlowTimeSampling, PRNGKey, TimeSamplingStrategy


class LossStrategy(ABC):
    @abstractmethod
    def compute_loss(self, state, key: PRNGKey, x: torch.Tensor, **kw):
        ...


def _prep_x(x):
    if x.dtype != torch.float32:
        x = x.float()
    return x.reshape(x.shape[0], -1).contiguous()


def _order_rows_plain_first(t, r, B, row0, Bg, prop, sampled, row_stride=1):
    """Permutation putting rows with r == t FIRST (the merged two-pass schedule of ``model.forward_imf``); returns
    (perm | None, n_tan).  With sampled times the r == t rows already are a local prefix (utils.py:41-44 applied per
    global row), so no permutation is needed at all."""
    if sampled:
        gsz = ops.data_size_of(Bg, prop)
        ds = min(max(-((row0 - gsz) // row_stride), 0), B)
        return None, B - ds
    mask = (t.reshape(-1) != r.reshape(-1))
    n_tan = int(mask.sum().item())
    if n_tan == B or n_tan == 0:
        return None, n_tan
    perm = torch.argsort(mask.to(torch.int8), stable=True)
    return (None if bool((perm == torch.arange(B, device=perm.device)).all()) else perm), n_tan


def _order_rows(t, r, B, row0, Bg, prop, sampled, row_stride=1):
    """Permutation putting rows with r != t first; returns (perm | None, n_tan)."""
    if sampled:
        # local rows i with row0 + i*row_stride < data_size have r == t: a local prefix [0, ds).  data_size is the
        # SAME integer the kernel received (ops.data_size_of; ADVICE r1: the kernel used to recompute it in float).
        gsz = ops.data_size_of(Bg, prop)
        ds = min(max(-((row0 - gsz) // row_stride), 0), B)   # ceil((gsz - row0) / row_stride), clamped
        if ds == 0:
            return None, B
        if ds == B:
            return None, 0
        perm = torch.cat([torch.arange(ds, B), torch.arange(0, ds)]).to(t.device)
        return perm, B - ds
    mask = (t.reshape(-1) != r.reshape(-1))
    n_tan = int(mask.sum().item())
    if n_tan == B or n_tan == 0:
        return None, n_tan
    perm = torch.argsort((~mask).to(torch.int8), stable=True)
    return perm, n_tan


def _loss_mode(use_weighted_loss):
    return 0 if use_weighted_loss else 1


class FlowMatchingLoss(LossStrategy):
    def __init__(self, noise_schedule: NoiseSchedule | None = None,
                 time_sampling: TimeSamplingStrategy | None = None, use_weighted_loss: bool = True):
        self.noise_schedule = noise_schedule or LinearNoiseSchedule()
        self.time_sampling = time_sampling or LogitNormalTimeSampling()
        self.use_weighted_loss = use_weighted_loss

    def compute_loss(self, state, key, x, *, e=None, t=None, row0=0, global_batch=None, aux=None, on_block=None,
                     fused=None, row_stride=1):
        model, w = state.model, state.work
        x = _prep_x(x)
        B = x.shape[0]
        Bg = global_batch or B
        if t is None:
            t = self.time_sampling.sample_time(key, B, row0=row0, global_batch=Bg, device=x.device,
                                               row_stride=row_stride)
        t = t.reshape(B, 1).float().contiguous()
        ns = self.noise_schedule
        z, target, _ = ops.flow_prepare(x, t, model.dtype, ns.noise_min, ns.noise_max, e=e, seed=key.seed,
                                        step=key.counter, row0=row0, row_stride=row_stride)
        ctx_holder = model.new_ctx()
        latents = model.encode(w, x, ctx_holder)
        cond, _ = model.conditioning(w, t, torch.zeros_like(t), latents)
        pred, _, ctx = model.forward(w, z, cond, latents=latents, save=True, ctx=ctx_holder)
        loss, du, _ = ops.flow_loss(pred, target, kind=0, mode=_loss_mode(self.use_weighted_loss), Bglobal=Bg)
        grads = state.grad_buffers()
        _, dcond, dlat = model.backward(w, ctx, du, grads, on_block=on_block, fused=fused)
        model.backward_conditioning(w, ctx, dcond, latents, grads, dlat=dlat)
        if aux is not None:
            aux.update(pred=pred, t=t)
        return loss, grads


class _TwoTimeLoss(LossStrategy):
    kind = 0

    def _run(self, state, key, x, e, t, r, row0, global_batch, aux, *, nmin, nmax, mode, p, c, use_v_pass,
             on_block=None, fused=None, row_stride=1, want_grads=True):
        model, w = state.model, state.work
        x = _prep_x(x)
        B = x.shape[0]
        Bg = global_batch or B
        sampled = t is None
        if sampled:
            t, r = self.time_sampling.sample_time_pair(key, B, row0=row0, global_batch=Bg, device=x.device,
                                                       row_stride=row_stride)
        t = t.reshape(B, 1).float().contiguous()
        r = r.reshape(B, 1).float().contiguous()
        prop = getattr(self.time_sampling, "data_proportion", 0.5)
        # Optional two-pass schedule (MFC_IMF_MERGE=1): measured on MI355X at the literal shape it is NOT faster than the
        # row-stacked default (161.9 vs 160.1 ms per step: two 128-row N-streaming products cost as much as a 64-row and
        # a 192-row one, and the ConvNeXt kernels are indifferent to the split), so it stays off; kept because it
        # halves the largest activation batch (2 x 128 rows instead of 64 + 192).
        if use_v_pass and hasattr(model, "forward_imf") and os.environ.get("MFC_IMF_MERGE", "0") == "1":
            _, n_tan0 = _order_rows_plain_first(t, r, B, row0, Bg, prop, sampled, row_stride)
            if 0 < n_tan0 < B:
                return self._run_merged(state, key, x, e, t, r, row0, Bg, aux, nmin=nmin, nmax=nmax, mode=mode, p=p, c=c,
                                        on_block=on_block, fused=fused, row_stride=row_stride, want_grads=want_grads,
                                        sampled=sampled, prop=prop)
        perm, n_tan = _order_rows(t, r, B, row0, Bg, prop, sampled, row_stride)
        prep = dict(seed=key.seed, step=key.counter)
        if perm is None:
            z, target, _ = ops.flow_prepare(x, t, model.dtype, nmin, nmax, e=e, row0=row0, row_stride=row_stride, **prep)
        elif e is not None:
            x, t, r, e = x[perm].contiguous(), t[perm].contiguous(), r[perm].contiguous(), e[perm].contiguous()
            z, target, _ = ops.flow_prepare(x, t, model.dtype, nmin, nmax, e=e, **prep)
        elif sampled:
            # the permutation is the rotation [ds, B) ++ [0, ds): two launches keep the Philox noise keyed by the
            # GLOBAL row each sample had before the re-ordering (shards then sum to the full batch exactly)
            ds = B - n_tan
            x, t, r = x[perm].contiguous(), t[perm].contiguous(), r[perm].contiguous()
            z = torch.empty((B, x.shape[1]), dtype=model.dtype, device=x.device)
            target = torch.empty((B, x.shape[1]), dtype=torch.float32, device=x.device)
            ops.flow_prepare(x[:n_tan], t[:n_tan], model.dtype, nmin, nmax, row0=row0 + ds * row_stride,
                             row_stride=row_stride, out=(z[:n_tan], target[:n_tan]), **prep)
            ops.flow_prepare(x[n_tan:], t[n_tan:], model.dtype, nmin, nmax, row0=row0, row_stride=row_stride,
                             out=(z[n_tan:], target[n_tan:]), **prep)
        else:
            # explicit (t, r) with an arbitrary r == t pattern and no explicit noise: draw in the original order, permute
            z, target, _ = ops.flow_prepare(x, t, model.dtype, nmin, nmax, row0=row0, row_stride=row_stride, **prep)
            x, t, r = x[perm].contiguous(), t[perm].contiguous(), r[perm].contiguous()
            z, target = z[perm].contiguous(), target[perm].contiguous()
        ctx_holder = model.new_ctx()
        latents = model.encode(w, x, ctx_holder)
        h = ops.axpby(1.0, t, -1.0, r)
        cond_u, cdot = model.conditioning(w, t, h, latents, want_dot=n_tan > 0)
        zdot = None
        if n_tan > 0:
            if use_v_pass:
                # boundary-condition velocity v = f(z, [t, 0]) on the rows that need a tangent
                lat_t = None if latents is None else latents[:n_tan]
                cond_v, _ = model.conditioning(w, t[:n_tan].contiguous(), torch.zeros_like(t[:n_tan]), lat_t)
                v, _, _ = model.forward(w, z[:n_tan], cond_v, latents=lat_t)
                zdot = v
            else:
                zdot = ops.cast(target[:n_tan].contiguous(), model.dtype)  # MeanFlow: tangent (e - x)
        u, dudt, ctx = model.forward(w, z, cond_u, xdot=zdot, cond_dot=cdot, latents=latents, save=True,
                                     ctx=ctx_holder)
        loss, du, pe = ops.flow_loss(u, target, dudt=dudt, n_tan=n_tan, t=t, r=r, kind=self.kind, mode=mode, p=p,
                                     c=c, Bglobal=Bg, want_grad=want_grads)
        grads = None
        if want_grads:          # (False: evaluation only -- loss and ``aux``, no reverse pass, no gradient buffers)
            grads = state.grad_buffers()
            _, dcond, dlat = model.backward(w, ctx, du, grads, on_block=on_block, fused=fused)
            model.backward_conditioning(w, ctx, dcond, latents, grads, dlat=dlat)
        if aux is not None:
            inv = None if perm is None else torch.argsort(perm)
            un = (lambda a: a) if inv is None else (lambda a: a[inv])
            aux.update(u=un(u), t=un(t), r=un(r), n_tan=n_tan, per_example=un(pe),
                       dudt=dudt, perm=perm, v=(zdot if use_v_pass else None))
        return loss, grads


    def _run_merged(self, state, key, x, e, t, r, row0, Bg, aux, *, nmin, nmax, mode, p, c, on_block, fused, row_stride,
                    want_grads, sampled, prop):
        """improved MeanFlow with the model's two-pass schedule (``forward_imf``): rows are kept [r == t rows; tangent
        rows] -- with sampled times that is the order the batch already has -- the boundary velocity pass carries the
        r == t rows of the u pass along, and the tangent pass runs on the tangent rows alone.  Same arithmetic per row
        as ``_run``; only the grouping of rows into launches differs."""
        model, w = state.model, state.work
        B = x.shape[0]
        perm, n_tan = _order_rows_plain_first(t, r, B, row0, Bg, prop, sampled, row_stride)
        n_plain = B - n_tan
        prep = dict(seed=key.seed, step=key.counter)
        z, target, _ = ops.flow_prepare(x, t, model.dtype, nmin, nmax, e=e, row0=row0, row_stride=row_stride, **prep)
        if perm is not None:     # explicit (t, r) with r == t rows not in front: noise was drawn in the original order
            x, t, r = x[perm].contiguous(), t[perm].contiguous(), r[perm].contiguous()
            z, target = z[perm].contiguous(), target[perm].contiguous()
        ctx_holder = model.new_ctx()
        latents = model.encode(w, x, ctx_holder)
        h = ops.axpby(1.0, t, -1.0, r)
        cond_u, cdot = model.conditioning(w, t, h, latents, want_dot=True)
        tt = t[n_plain:].contiguous()
        lat_t = None if latents is None else latents[n_plain:]
        cond_v, _ = model.conditioning(w, tt, torch.zeros_like(tt), lat_t)       # boundary velocity v = f(z, [t, 0])
        u, dudt, v, ctx = model.forward_imf(w, z, cond_u, cond_v, cdot[n_plain:].contiguous(), n_plain, ctx=ctx_holder)
        loss, du, pe = ops.flow_loss(u, target, dudt=dudt, n_tan=-n_tan, t=t, r=r, kind=self.kind, mode=mode, p=p, c=c,
                                     Bglobal=Bg, want_grad=want_grads)
        grads = None
        if want_grads:
            grads = state.grad_buffers()
            _, dcond, dlat = model.backward(w, ctx, du, grads, on_block=on_block, fused=fused)
            model.backward_conditioning(w, ctx, dcond, latents, grads, dlat=dlat)
        if aux is not None:
            inv = None if perm is None else torch.argsort(perm)
            un = (lambda a: a) if inv is None else (lambda a: a[inv])
            # ``dudt`` / ``v`` rows: the tangent rows in batch order (= rows n_plain.. of the permuted batch)
            aux.update(u=un(u), t=un(t), r=un(r), n_tan=n_tan, per_example=un(pe), dudt=dudt, perm=perm, v=v,
                       tangent_rows_last=True)
        return loss, grads


class MeanFlowLoss(_TwoTimeLoss):
    """trainers/loss_strategies.py:115-201 (uniform interpolation regardless of noise_schedule, :156-160)."""
    kind = 1

    def __init__(self, noise_schedule: NoiseSchedule | None = None,
                 time_sampling: MeanFlowTimeSampling | None = None, gamma: float = 0.5, c: float = 1e-3):
        self.noise_schedule = noise_schedule or LinearNoiseSchedule()
        self.time_sampling = time_sampling or MeanFlowTimeSampling()
        self.gamma = gamma
        self.c = c

    def compute_loss(self, state, key, x, *, e=None, t=None, r=None, row0=0, global_batch=None, aux=None,
                     on_block=None, fused=None, row_stride=1, want_grads=True):
        return self._run(state, key, x, e, t, r, row0, global_batch, aux, nmin=0.0, nmax=1.0, mode=2,
                         p=1.0 - self.gamma, c=self.c, use_v_pass=False, on_block=on_block, fused=fused,
                         row_stride=row_stride, want_grads=want_grads)


class ImprovedMeanFlowLoss(_TwoTimeLoss):
    """trainers/loss_strategies.py:204-280."""
    kind = 0

    def __init__(self, noise_schedule: NoiseSchedule | None = None,
                 time_sampling: MeanFlowTimeSampling | None = None, use_weighted_loss: bool = True):
        self.noise_schedule = noise_schedule or LinearNoiseSchedule()
        self.time_sampling = time_sampling or MeanFlowTimeSampling()
        self.use_weighted_loss = use_weighted_loss

    def compute_loss(self, state, key, x, *, e=None, t=None, r=None, row0=0, global_batch=None, aux=None,
                     on_block=None, fused=None, row_stride=1, want_grads=True):
        ns = self.noise_schedule
        return self._run(state, key, x, e, t, r, row0, global_batch, aux, nmin=ns.noise_min, nmax=ns.noise_max,
                         mode=_loss_mode(self.use_weighted_loss), p=1.0, c=1e-3, use_v_pass=True, on_block=on_block,
                         fused=fused, row_stride=row_stride, want_grads=want_grads)
