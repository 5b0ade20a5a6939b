"""Data-parallel gradient exchange: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL
over xGMI on ROCm; "gloo" on CPU for tests).

The reference has no distributed path: this is new.  The batch is sharded across ranks
(rank k owns global rows [k*B, (k+1)*B)); every loss already divides by the GLOBAL batch, so the
exchange is a plain SUM all-reduce of the gradient buffers between ``compute_loss`` and
``apply_gradients`` (trainers/training_steps.py:32-33 is where it slots in).  Big kernels are reduced
in place tensor-by-tensor (each is its own multi-GB bucket, launched asynchronously so RCCL pipelines
them); everything small is packed into one fp32 bucket.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class GradReducer:
    def __init__(self, group=None, small_numel: int = 1 << 20):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.small_numel = small_numel
        self.world = dist.get_world_size(group)
        self._flat = None

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
