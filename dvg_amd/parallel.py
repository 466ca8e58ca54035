"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm)
on xGMI, gloo on CPU for the tests.  The reference has no distributed code at all;
this is the MI355X-side addition of §8(e).

Gradients are averaged with ONE all-reduce of a flat fp32 buffer per backward pass (vgg_64: 21.1 M
parameters = 84.6 MB; dcgan_64: 44 MB).  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a single
large message lets RCCL use all links at once, whereas per-parameter all-reduces (170 tensors, many of
them 64-512 floats) would be latency-bound.  BatchNorm uses per-replica statistics (DDP semantics).
"""
from __future__ import annotations

import os
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> tuple:
    """Returns (rank, world, local_rank); initialises the default process group when WORLD_SIZE > 1."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            torch.cuda.set_device(local)
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_batch(global_batch: int, world: int) -> int:
    if global_batch % world:
        raise ValueError(f"global batch {global_batch} is not divisible by {world} ranks")
    return global_batch // world


class FlatGradReducer:
    """Average the gradients of `params` across ranks with one all-reduce of a flat buffer."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        self._flat = None

    def _buffer(self, like: torch.Tensor) -> torch.Tensor:
        if self._flat is None or self._flat.device != like.device:
            self._flat = torch.zeros(self.numel, device=like.device, dtype=torch.float32)
        return self._flat

    @torch.no_grad()
    def reduce(self) -> None:
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1 or not self.params:
            return
        flat = self._buffer(self.params[0])
        off = 0
        for p in self.params:  # parameters a pass did not touch contribute zeros (still averaged)
            n = p.numel()
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.div_(dist.get_world_size(self.group))
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is not None:
                p.grad.copy_(flat[off:off + n].view_as(p.grad))
            off += n


def broadcast_parameters(modules: Iterable[torch.nn.Module], src: int = 0, group=None) -> None:
    """Make every replica start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for m in modules:
            for t in list(m.parameters()) + list(m.buffers()):
                dist.broadcast(t, src=src, group=group)
