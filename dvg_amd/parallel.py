"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL ("nccl" backend on ROCm)
on xGMI, gloo on CPU for the tests.  The reference has no distributed code at all;
this is the MI355X-side addition of §8(e).

Gradients live in ONE flat fp32 arena (dvg_amd/optim.py: `p.grad` of every parameter is a view of it, and the
fused Adam kernel reads the same buffer), so averaging them across ranks is an all-reduce of a contiguous RANGE
of that buffer, in place: no gather into a communication buffer, no scatter back, no copy into the optimiser
(vgg_64: 21.1 M parameters = 84.6 MB; dcgan_64: 44 MB).  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a
few large messages let RCCL use all links at once, whereas per-parameter all-reduces (170 tensors, many of them
64-512 floats) would be latency-bound.  `ArenaReducer.start()` issues the collective asynchronously (RCCL runs it
on its own stream, ordered after the work already queued on the caller's stream) and `finish()` makes the caller's
stream wait for it: train.Trainer starts the decoder / LSTM / GP range as soon as the decoder phase of the backward
pass is done and the encoder range after the encoder phase, so the first - larger - collective overlaps the encoder
backward.  BatchNorm uses per-replica statistics (DDP semantics).
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
    if os.environ.get("DVG_DP_SHARE_GPU") == "1":
        local = 0      # rehearsal of the multi-rank control flow on a one-GPU box: every rank on device 0 (use with gloo)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if backend is None:
            backend = os.environ.get("DVG_DP_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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


def _active(group) -> bool:
    return dist.is_initialized() and (dist.get_world_size(group) > 1 or os.environ.get("DVG_FORCE_ALLREDUCE") == "1")


class ArenaReducer:
    """Average ranges of a flat gradient buffer across ranks, in place and asynchronously."""

    def __init__(self, flat_grad: torch.Tensor, group=None):
        self.g, self.group = flat_grad, group
        self.enabled = True
        self.calls = 0          # collectives issued (bench.py reports it)
        self.floats = 0

    def active(self) -> bool:
        return self.enabled and _active(self.group)

    def start(self, lo: int, hi: int):
        """Issue the all-reduce of g[lo:hi]; returns a handle for finish() (None when there is nothing to do)."""
        if not self.active() or hi <= lo:
            return None
        buf = self.g[lo:hi]
        world = dist.get_world_size(self.group)
        self.calls += 1
        self.floats += hi - lo
        if dist.get_backend(self.group) == "nccl":
            return (dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=True), None)
        if buf.is_cuda:
            # gloo on DEVICE memory (the one-GPU rehearsal: both ranks on GPU 0): an explicit, synchronous host round trip rather
            # than gloo's own CUDA path (worker threads, side streams) - a suspect removed while bisecting the rehearsal's
            # run-to-run differences (profiles/r06_dp_race_bisect.txt; it was not the cause).  The rehearsal measures nothing, so
            # nothing is lost; RCCL (above) is the path that overlaps.
            host = buf.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            buf.copy_(host.div_(world))
            return None
        return (dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True), (buf, world))

    @staticmethod
    def finish(handle) -> None:
        if handle is None:
            return
        work, post = handle
        work.wait()             # nccl: the current stream waits (the host does not); gloo: blocks
        if post is not None:
            post[0].div_(post[1])

    def reduce(self, lo: int, hi: int) -> None:
        self.finish(self.start(lo, hi))


class FlatGradReducer:
    """Average the gradients of an arbitrary parameter list with one all-reduce of a gathered flat buffer (for
    parameters that do not live in a FusedAdam arena; train.Trainer uses ArenaReducer)."""

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
        if not _active(self.group) or not self.params:
            return
        flat = self._buffer(self.params[0])
        views, grads, off = [], [], 0
        for p in self.params:  # parameters a pass did not touch contribute zeros (still averaged)
            n = p.numel()
            v = flat[off:off + n].view(p.shape)
            if p.grad is None:
                v.zero_()
            else:
                views.append(v)
                grads.append(p.grad)
            off += n
        if views:
            torch._foreach_copy_(views, grads)
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
        flat.div_(dist.get_world_size(self.group))
        if views:
            torch._foreach_copy_(grads, views)


def broadcast_coalesced(tensors: Iterable[torch.Tensor], src: int = 0, group=None) -> int:
    """Broadcast rank `src`'s values of `tensors` with ONE collective per dtype: the tensors are packed into a flat buffer,
    broadcast, and copied back in place.  Returns the number of collectives issued."""
    by_dtype = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    n = 0
    with torch.no_grad():
        for ts in by_dtype.values():
            flat = torch.cat([t.reshape(-1) for t in ts])
            dist.broadcast(flat, src=src, group=group)
            n += 1
            off = 0
            for t in ts:
                t.copy_(flat[off:off + t.numel()].view(t.shape))
                off += t.numel()
    return n


def broadcast_parameters(modules: Iterable[torch.nn.Module], src: int = 0, group=None, arena_p: Optional[torch.Tensor] = None) -> int:
    """Make every replica start from rank `src`'s parameters and buffers.  With `arena_p` (the flat parameter buffer of
    optim.FlatArena, of which every parameter of `modules` is a view) the parameters go as ONE broadcast of that buffer,
    in place; the buffers (BatchNorm running statistics, step counters) and any parameter outside the arena are packed per
    dtype.  xGMI is point-to-point: ~200 small broadcasts cost ~200 latencies, two or three large ones use all links.
    Returns the number of collectives issued (0 with one rank)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return 0
    modules = list(modules)
    n, rest = 0, []
    with torch.no_grad():
        if arena_p is not None:
            dist.broadcast(arena_p, src=src, group=group)
            n += 1
            lo = arena_p.data_ptr()
            hi = lo + arena_p.numel() * arena_p.element_size()
            rest = [p for m in modules for p in m.parameters() if not (lo <= p.data_ptr() < hi)]
        else:
            rest = [p for m in modules for p in m.parameters()]
        rest += [b for m in modules for b in m.buffers()]
        if rest:
            n += broadcast_coalesced(rest, src=src, group=group)
    return n
