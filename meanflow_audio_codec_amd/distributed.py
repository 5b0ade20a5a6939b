"""Data-parallel gradient exchange: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

The reference has no distributed path: this is new.  The batch is sharded across ranks,
INTERLEAVED: rank k of G owns global rows k, k+G, k+2G, ... (``shard_rows``), so that the deterministic
"first int(B*p) global rows have r = t" rule (utils.py:41-44) gives every rank the same share (+-1) of r == t
rows (3 forward-equivalents of work) and of tangent rows (5) -- contiguous ownership would hand the first half
of the ranks only cheap rows and make every step wait for the other half.  Every loss already divides by the GLOBAL batch, so the
exchange is a plain SUM all-reduce of the gradient buffers between ``compute_loss`` and
``apply_gradients`` (trainers/training_steps.py:32-33 is where it slots in).  Big kernels are reduced
in place tensor-by-tensor (each is its own multi-GB bucket, launched asynchronously so RCCL pipelines
them); everything small is packed into one fp32 bucket.

Sharded optimizer (``shard_optimizer=True``, the default for more than one rank): for the big bf16-stored kernels the
all-reduce is split into its two halves around the optimizer -- reduce-scatter of the gradient, AdamW on this rank's
1/world slice of (master, m, v), all-gather of the updated bf16 working copy.  Same bytes on the links as the
all-reduce, but each GPU streams only 1/world of the 28 B/parameter optimizer traffic (57 of the 65 ms of ``mfc_adamw``
at 8 GPUs).  A rank's fp32 master and moments are then authoritative only on its own slice; ``gather_master`` restores
the full tensors everywhere (before a checkpoint).  gloo has no reduce-scatter: there the same slices are produced
with an all-reduce and ``all_gather`` on views, so the CPU / single-GPU rehearsals run the same index arithmetic.

Deferred gather (``defer_gather``, default on): the links are busy with the reduce-scatters for most of the reverse
pass, so the all-gathers are not interleaved with them but issued together once the reverse pass is done, first block
first; each leaves an event in ``state.work.pending`` and the next step's forward waits per leaf, at its first read
(``models/train_state.py::WorkDict``).  The all-gather phase then overlaps the next forward instead of sitting between
two steps; AdamW writes its bf16 slice straight into the working copy and the all-gather runs in place on it.

Precision of the exchange: the big kernels' gradients are stored, and summed across ranks, in bf16 (RCCL adds in
fp32 inside one reduction step but every hop of the ring re-rounds the running sum to bf16).  With u = 2^-8 (half a
bf16 ulp) and A = sum over ranks of |contribution|: worst case (G - 1) u A per element, RMS about sqrt(G - 1) times the
single rounding an fp32 exchange into the same bf16 buffer would make -- the same order as the bf16 rounding of each
rank's own contribution (``tests/test_bench_launcher.py::test_bf16_exchange_error_bound`` measures both exchanges
against the exact sum at G = 4 and 8).  The AdamW update normalises gradient magnitudes, so this noise moves an update
by that fraction of lr.  Small leaves and the ConvNeXt interior's gradients are exchanged in fp32.

Which collective is used (native ``reduce_scatter_tensor`` / ``all_gather_into_tensor`` or their all-reduce /
all-gather emulation) is decided ONCE in ``__init__`` from the backend and the environment alone -- no collective is
involved in the decision, so it cannot itself desynchronise the ranks -- and every rank then issues the same sequence
of collectives.  Errors propagate (a rank that fails exits non-zero and the launcher tears the job down) instead of
silently switching one rank to a different collective, which would hang its peers.  Constructing a ``GradReducer`` is
therefore NOT a collective; using it is (every rank must make the same calls in the same order).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def shard_rows(rank: int, world: int, per_rank_batch: int) -> dict:
    """Row ownership of one rank under the interleaved layout: keyword arguments for ``train_step`` /
    ``compute_loss`` (``row0``, ``row_stride``, ``global_batch``).  Local row i is global row rank + i * world."""
    return dict(row0=int(rank), row_stride=int(world), global_batch=int(world) * int(per_rank_batch))


def shard_of(x: torch.Tensor, rank: int, world: int) -> torch.Tensor:
    """This rank's rows of a global batch ``x`` [G*B, ...] under the interleaved layout."""
    return x[rank::world].contiguous()


class GradReducer:
    def __init__(self, group=None, small_numel: int = 1 << 20, shard_optimizer: bool | None = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.small_numel = small_numel
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self._flat = None
        if shard_optimizer is None:
            shard_optimizer = os.environ.get("MFC_SHARD_OPTIMIZER", "1") != "0"
        self.shard_optimizer = bool(shard_optimizer) and self.world > 1
        self.defer_gather = self.shard_optimizer and os.environ.get("MFC_DEFER_GATHER", "1") != "0"
        self._tmp = {}
        self.sharded: set = set()                            # leaves whose master / moments live in slices
        self._native = self._decide_native()                 # reduce_scatter_tensor / all_gather_into_tensor

    def _decide_native(self) -> bool:
        """Taken once, identically on every rank and WITHOUT a collective: a pure function of the backend (RCCL has
        ``reduce_scatter_tensor`` / ``all_gather_into_tensor``; gloo has no reduce-scatter and takes the emulation) and
        of ``MFC_DIST_NATIVE=0/1``, which must be set alike on all ranks (the launcher hands one environment to every
        rank).  There is deliberately no probe: a probe collective that raises on one rank only would leave that rank
        in a different collective from its peers (a hang).  If a native collective fails later, the error propagates,
        the rank exits non-zero and the launcher tears the job down."""
        env = os.environ.get("MFC_DIST_NATIVE")
        native = (dist.get_backend(self.group) == "nccl") if env is None else (env == "1")
        return bool(native) and self.world > 1

    # ---- sharded optimizer ------------------------------------------------------------------------------------
    def _buf(self, key, n, dtype, device):
        t = self._tmp.get(key)
        if t is None or t.numel() != n or t.dtype != dtype or t.device != device:
            t = torch.empty(n, dtype=dtype, device=device)
            self._tmp[key] = t
        return t

    def _reduce_scatter(self, out: torch.Tensor, full: torch.Tensor) -> None:
        """out <- this rank's slice of the SUM over ranks of ``full`` (flat, numel = world * out.numel()).
        The collective was chosen in ``__init__``; an error here propagates (no per-call fallback: see module doc)."""
        if self._native:
            dist.reduce_scatter_tensor(out, full, op=dist.ReduceOp.SUM, group=self.group)
            return
        dist.all_reduce(full, op=dist.ReduceOp.SUM, group=self.group)
        out.copy_(full[self.rank * out.numel():(self.rank + 1) * out.numel()])

    def _all_gather(self, full: torch.Tensor, mine: torch.Tensor) -> None:
        """full (flat) <- concatenation over ranks of ``mine``; ``mine`` may be this rank's slice of ``full`` (in place)."""
        if self._native:
            dist.all_gather_into_tensor(full, mine, group=self.group)
            return
        n = mine.numel()
        if mine.data_ptr() == full[self.rank * n:(self.rank + 1) * n].data_ptr():
            mine = mine.clone()
        dist.all_gather([full[r * n:(r + 1) * n] for r in range(self.world)], mine, group=self.group)

    def shardable(self, state, name: str, grad: torch.Tensor) -> bool:
        w = dict.get(state.work, name)          # raw access: never consume a pending-gather event on this stream
        return (self.shard_optimizer and w is not None and w.dtype == torch.bfloat16 and grad.dtype == torch.bfloat16
                and grad.is_contiguous() and w.is_contiguous() and grad.numel() > self.small_numel
                and grad.numel() % (self.world * 64) == 0)

    def sharded_update(self, state, names, grads: dict, defer: list | None = None) -> list:
        """One optimizer step (``state.step`` already advanced by ``begin_update``) of every shardable leaf in ``names``:
        reduce-scatter the gradient, AdamW on the own slice (the bf16 result goes straight into this rank's slice of
        the working copy), all-gather the working copy in place -- now, or, with ``defer`` (a list the names are
        appended to), later in ``flush_gathers``.  Returns the names that were NOT handled (small / fp32 leaves:
        all-reduce + full AdamW, the caller's job)."""
        from . import ops
        rest = []
        tx = state.tx
        for k in names:
            g = grads[k]
            if not self.shardable(state, k, g):
                rest.append(k)
                continue
            n = g.numel()
            sh = n // self.world
            lo = self.rank * sh
            gsh = self._buf(("g", sh), sh, torch.bfloat16, g.device)
            self._reduce_scatter(gsh, g.view(-1))
            wflat = dict.__getitem__(state.work, k).view(-1)
            ops.adamw(state.params[k].view(-1)[lo:lo + sh], gsh, state.opt_state["mu"][k].view(-1)[lo:lo + sh],
                      state.opt_state["nu"][k].view(-1)[lo:lo + sh], lr=tx.learning_rate, wd=tx.weight_decay,
                      step=state.step, b1=tx.b1, b2=tx.b2, eps=tx.eps, p_bf16=wflat[lo:lo + sh])
            if defer is None:
                self._all_gather(wflat, wflat[lo:lo + sh])
            else:
                defer.append(k)
            self.sharded.add(k)
        return rest

    @staticmethod
    def _use_order(name: str):
        """Order in which a forward pass first reads the big kernels: block by block, and inside a block
        input_proj1 -> input_proj2 -> output_proj1 -> output_proj2 (models/conv_flow.py); other names keep their order."""
        parts = name.split("/")
        blk = int(parts[0].split("_")[1]) if parts[0].startswith("blocks_") and parts[0].split("_")[1].isdigit() else 1 << 30
        within = {"input_proj1": 0, "input_proj2": 1, "output_proj1": 2, "output_proj2": 3}.get(
            parts[1] if len(parts) > 1 else "", 4)
        return (blk, within)

    def flush_gathers(self, state, names: list) -> None:
        """All-gather (in place) the working copy of every deferred leaf on the CURRENT stream, in the order the next
        forward reads them; each leaf gets an event in ``state.work.pending``."""
        for k in sorted(names, key=self._use_order):
            wflat = dict.__getitem__(state.work, k).view(-1)
            sh = wflat.numel() // self.world
            self._all_gather(wflat, wflat[self.rank * sh:(self.rank + 1) * sh])
            if wflat.is_cuda and hasattr(state.work, "pending"):
                ev = torch.cuda.Event()
                ev.record()
                state.work.pending[k] = ev
        del names[:]

    def gather_master(self, state) -> None:
        """Make the fp32 master and both moments of every sharded leaf complete on every rank (checkpointing)."""
        getattr(state.work, "wait_all", lambda: None)()
        for k in sorted(self.sharded):
            for t in (state.params[k], state.opt_state["mu"][k], state.opt_state["nu"][k]):
                flat = t.view(-1)
                sh = flat.numel() // self.world
                mine = flat[self.rank * sh:(self.rank + 1) * sh].clone()
                self._all_gather(flat, mine)

    def reduce_tensors(self, tensors) -> None:
        """SUM all-reduce a list of gradient tensors in place on the CURRENT stream (one block of the
        reverse pass): big ones individually, the small fp32 ones packed into one flat bucket."""
        if self.world == 1:
            return
        small = [t for t in tensors if t.numel() <= self.small_numel and t.dtype == torch.float32]
        ids = {id(t) for t in small}
        for t in tensors:
            if id(t) not in ids:
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        if small:
            flat = torch.cat([t.reshape(-1) for t in small])
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            off = 0
            for t in small:
                t.reshape(-1).copy_(flat[off:off + t.numel()])
                off += t.numel()

    def reduce_scalar(self, loss: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return loss
        out = loss.detach().clone().to(torch.float32)
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.group)
        return out

    def reduce(self, grads: dict, loss: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return loss
        handles = []
        small = [k for k, g in grads.items() if g.numel() <= self.small_numel and g.dtype == torch.float32]
        big = [k for k in grads if k not in set(small)]
        for k in big:
            handles.append(dist.all_reduce(grads[k], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        n = sum(grads[k].numel() for k in small) + 1
        if self._flat is None or self._flat.numel() != n or self._flat.device != loss.device:
            self._flat = torch.empty(n, dtype=torch.float32, device=loss.device)
        off = 0
        for k in small:
            m = grads[k].numel()
            self._flat[off:off + m].copy_(grads[k].reshape(-1))
            off += m
        self._flat[off] = loss.to(torch.float32)
        handles.append(dist.all_reduce(self._flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for h in handles:
            h.wait()
        off = 0
        for k in small:
            m = grads[k].numel()
            grads[k].reshape(-1).copy_(self._flat[off:off + m])
            off += m
        return self._flat[off].clone()
