"""TrainState + AdamW -- host mirror of ``models/train_state.py`` (a flax ``TrainState``) and of the
``optax.adamw`` transformation the reference builds at ``trainers/train.py:236``.

The update itself is the fused HIP kernel ``mfc_adamw`` (fp32 master, fp32 moments, optional bf16
working copy).  Unlike flax the state is updated IN PLACE (13.7 B parameters cannot be copied per
step); ``apply_gradients`` returns ``self`` so reference-style code ``state = state.apply_gradients(..)``
keeps working.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

import torch

from .. import ops


@dataclass
class AdamW:
    """optax.adamw(learning_rate, weight_decay): b1=.9, b2=.999, eps=1e-8, decay on ALL leaves."""
    learning_rate: float
    weight_decay: float = 1e-4
    b1: float = 0.9
    b2: float = 0.999
    eps: float = 1e-8


def adamw(learning_rate: float, weight_decay: float = 1e-4, b1: float = 0.9, b2: float = 0.999,
          eps: float = 1e-8) -> AdamW:
    return AdamW(learning_rate, weight_decay, b1, b2, eps)


class WorkDict(dict):
    """The working copies the kernels read, keyed by leaf name.  A leaf whose all-gather from the previous optimizer
    step is still in flight on the side stream (``distributed.GradReducer.flush_gathers``) carries an event in
    ``pending``; the first ``w[name]`` / ``w.get(name)`` makes the CURRENT stream wait for it, so the next forward pass
    starts while later blocks are still being gathered.  ``wait_all`` does so for everything (graph capture,
    ``refresh_work``, anything that walks ``values()``)."""

    def __init__(self):
        super().__init__()
        self.pending: dict = {}

    def _arrive(self, k):
        ev = self.pending.pop(k, None)
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)

    def __getitem__(self, k):
        if self.pending:
            self._arrive(k)
        return dict.__getitem__(self, k)

    def get(self, k, default=None):
        if self.pending:
            self._arrive(k)
        return dict.get(self, k, default)

    def wait_all(self):
        for k in list(self.pending):
            self._arrive(k)


class TrainState:
    def __init__(self, apply_fn, params: dict, tx: AdamW, model=None, step: int = 0, opt_state=None):
        self.apply_fn = apply_fn
        self.params = params            # fp32 masters
        self.tx = tx
        self.model = model
        self.step = step
        self.opt_state = opt_state or {
            "mu": {k: torch.zeros_like(v) for k, v in params.items()},
            "nu": {k: torch.zeros_like(v) for k, v in params.items()},
        }
        self.work = WorkDict()          # what the kernels read
        self._grads = None
        self.refresh_work()

    @classmethod
    def create(cls, *, apply_fn, params, tx, model=None):
        return cls(apply_fn, params, tx, model=model)

    def _work_dtype(self, name):
        if self.model is not None and hasattr(self.model, "compute_dtype_of"):
            return self.model.compute_dtype_of(name)
        return torch.float32

    def refresh_work(self):
        self.work.wait_all()
        for k, p in self.params.items():
            dt = self._work_dtype(k)
            if dt == torch.float32:
                self.work[k] = p
            else:
                w = self.work.get(k)
                if w is None or w.dtype != dt or w.shape != p.shape:
                    w = torch.empty(p.shape, dtype=dt, device=p.device)
                    self.work[k] = w
                ops.cast(p, dt, out=w)

    def grad_buffers(self) -> dict:
        """Persistent gradient buffers: big kernels in the compute dtype (written once per step by the
        weight-gradient GEMM), everything else fp32.  Buffers are created on first access (``LazyGrads``): in the
        fused single-GPU schedule the big kernels' gradients are consumed inside the weight-gradient GEMM and never
        exist, so their 2 bytes per parameter (27 GB at the literal shape) are never allocated."""
        if self._grads is None:
            self._grads = LazyGrads(self)
        return self._grads

    def reinit(self, seed: int = 0, tx: AdamW | None = None):
        """Back to a freshly initialised state IN PLACE (no second copy of a 13.7 B-parameter tree): every leaf is
        re-drawn by the model's own initialiser one leaf at a time, the moments are zeroed, the step counter reset and
        the working copies refreshed."""
        fresh_gen = getattr(self.model, "init_into", None)
        if fresh_gen is not None:
            fresh_gen(self.params, seed)
        else:
            from .common import init_into
            init_into(self.params, self.model.param_shapes(), seed)
        for d in (self.opt_state["mu"], self.opt_state["nu"]):
            for v in d.values():
                v.zero_()
        self.step = 0
        if tx is not None:
            self.tx = tx
        self.refresh_work()
        return self

    def fused_updater(self, names=None):
        """For a step opened with ``begin_update``: an object whose ``dw(name, A, dY, alpha)`` computes the weight
        gradient ``A^T dY`` of leaf ``name`` and applies its AdamW update in the same kernel (``mfc_gemm_adamw``).
        Only bf16-stored 2-D leaves with a 16-aligned second dimension qualify; for the others ``dw`` returns
        False and the caller materialises the gradient as usual."""
        return FusedUpdater(self, names)

    def begin_update(self):
        """Start one optimizer step whose leaves are updated piecewise with ``apply_subset``."""
        self.step += 1

    # leaves below this many elements share launches (mfc_adamw_multi); larger ones take the vectorised mfc_adamw
    MULTI_TENSOR_BELOW = 1 << 20

    def apply_subset(self, names, grads: dict, grad_scale: float = 1.0):
        """AdamW on the named leaves (``self.step`` is the 1-based update count, already advanced by the caller).  On
        the GPU the small leaves -- biases, conv kernels, GRN / layer-scale vectors: ~170 of a ConvFlow's 200 -- go
        through one ``mfc_adamw_multi`` launch per 48 instead of a 6 us launch each; the descriptor table of a
        (names, gradient buffers) combination is built once and reused (every tensor in it is updated in place)."""
        tx = self.tx
        hp = dict(lr=tx.learning_rate, wd=tx.weight_decay, step=self.step, b1=tx.b1, b2=tx.b2, eps=tx.eps,
                  grad_scale=grad_scale)
        names = list(names)
        small = [k for k in names if self.params[k].is_cuda and self.params[k].numel() < self.MULTI_TENSOR_BELOW]
        if len(small) > 1:
            mu, nu = self.opt_state["mu"], self.opt_state["nu"]
            # every pointer the descriptor table holds is part of the key (a reallocated working copy or moment must
            # never be updated through a stale table); dict.__getitem__: the key must not consume a pending-gather event
            key = (tuple(small), tuple((grads[k].data_ptr(), self.params[k].data_ptr(), mu[k].data_ptr(), nu[k].data_ptr(),
                                        dict.__getitem__(self.work, k).data_ptr()) for k in small))
            cache = self.__dict__.setdefault("_multi_cache", {})
            ent = cache.get(key)
            if ent is not None and self.work.pending:
                for k in small:          # a cached table still has to wait for in-flight all-gathers of these leaves
                    self.work._arrive(k)
            if ent is None:
                leaves = [(self.params[k], grads[k], self.opt_state["mu"][k], self.opt_state["nu"][k],
                           self.work[k] if self.work[k].dtype == torch.bfloat16 else None) for k in small]
                if len(cache) > max(16, getattr(self.model, "num_blocks", 0) + 2):
                    cache.clear()
                ent = cache[key] = (ops.adamw_multi_items(leaves), leaves)      # the tensors stay alive with the table
            (items, n), _ = ent
            ops.adamw_multi(items, n, **hp)
            small = set(small)
            names = [k for k in names if k not in small]
        for k in names:
            p, g, w = self.params[k], grads[k], self.work[k]
            ops.adamw(p, g, self.opt_state["mu"][k], self.opt_state["nu"][k],
                      p_bf16=(w if w.dtype == torch.bfloat16 else None), **hp)

    def apply_gradients(self, *, grads: dict, grad_scale: float = 1.0):
        self.step += 1
        self.apply_subset(list(self.params), grads, grad_scale)
        return self


class LazyGrads(dict):
    """name -> gradient buffer, allocated (zero-filled) on first access.  Iteration, ``in`` and ``len`` cover every
    parameter name whether or not its buffer exists yet."""

    def __init__(self, state: TrainState):
        super().__init__()
        self._state = state

    def __missing__(self, k):
        st = self._state
        p = st.params[k]                       # KeyError for an unknown name, like a plain dict
        dt = st._work_dtype(k)
        if "/conv_block/" in k:
            dt = torch.float32                 # fp32 accumulators (+=) of the spatial kernels' fixed-order reductions
        t = torch.zeros(p.shape, dtype=dt, device=p.device)
        dict.__setitem__(self, k, t)
        return t

    def get(self, k, default=None):
        return self[k] if k in self._state.params else default

    def __contains__(self, k):
        return k in self._state.params

    def __iter__(self):
        return iter(self._state.params)

    def __len__(self):
        return len(self._state.params)

    def keys(self):
        return self._state.params.keys()

    def items(self):
        return ((k, self[k]) for k in self._state.params)

    def values(self):
        return (self[k] for k in self._state.params)

    def allocated(self):
        """the buffers that exist (dict view, no allocation)"""
        return dict(dict.items(self))


class FusedUpdater:
    def __init__(self, state: TrainState, names=None):
        self.state = state
        self.names = None if names is None else set(names)
        self.done: set = set()

    def wants(self, name: str) -> bool:
        st = self.state
        if self.names is not None and name not in self.names:
            return False
        w = st.work.get(name)
        return (w is not None and w.dtype == torch.bfloat16 and w.dim() == 2 and w.shape[1] % 16 == 0
                and name not in self.done)

    def dw(self, name: str, A: torch.Tensor, dY: torch.Tensor, alpha: float = 1.0, bias_out: torch.Tensor | None = None,
           bias_scale: float = 1.0) -> bool:
        """``bias_out``: also write the bias gradient of the same Dense layer, ``bias_scale`` x the column sums of
        ``dY``, from the operand tiles the product stages (needs at least 33 rows: the 64-deep K-step kernel)."""
        if not self.wants(name) or A.dtype != torch.bfloat16 or dY.dtype != torch.bfloat16:
            return False
        if bias_out is not None and (dY.shape[0] <= 32 or os.environ.get("MFC_BIAS_FOLD", "1") == "0"):
            return False
        st, tx = self.state, self.state.tx
        ops.gemm_adamw(A, dY, trans_a=True, grad_scale=alpha, p=st.params[name], m=st.opt_state["mu"][name],
                       v=st.opt_state["nu"][name], p_bf16=st.work[name], lr=tx.learning_rate, wd=tx.weight_decay,
                       step=st.step, b1=tx.b1, b2=tx.b2, eps=tx.eps, colsum=bias_out, colsum_scale=bias_scale)
        self.done.add(name)
        return True
