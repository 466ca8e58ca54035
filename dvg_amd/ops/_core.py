"""Plumbing shared by the wrappers: the current HIP stream, the optional per-launch event timer (bench.py's roofline leg),
raw pointers, the NHWC-in-memory tensor helpers and the activation / mode codes of include/dvg_hip.h."""
from __future__ import annotations

import torch

from .._lib import check, lib


ACT_NONE, ACT_LRELU, ACT_TANH, ACT_SIGMOID = 0, 1, 2, 3


MODE_CONV3, MODE_CONV4S2, MODE_CONVT4S2 = 0, 1, 2


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class KernelTimer:
    """Optional per-launch HIP-event timing (bench.py's roofline leg).  Events are recorded on the
    stream the kernels are launched on (torch's current stream)."""

    def __init__(self):
        self.records = []  # (name, flops, bytes, ev0, ev1)

    def summary(self):
        """Per kernel name: launches, ms, EXECUTED flops, algorithmic bytes, and `alg_flops` = the direct-form FLOPs of the
        layer a launch belongs to (differs from `flops` only for the Winograd GEMMs, which execute 1/2.25 or 1/4 of them)."""
        torch.cuda.synchronize()
        agg = {}
        for name, fl, by, e0, e1, alg in self.records:
            a = agg.setdefault(name, {"launches": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "alg_flops": 0.0})
            a["launches"] += 1
            a["ms"] += e0.elapsed_time(e1)
            a["flops"] += fl
            a["bytes"] += by
            a["alg_flops"] += fl if alg is None else alg
        return agg


_timer = None


def set_timer(t):
    global _timer
    _timer = t


def _run(name, flops, nbytes, fn, *args, alg_flops=None):
    """Launch through the C ABI; with a KernelTimer installed, bracket the launch with events."""
    if _timer is None:
        check(fn(*args), name)
        return
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    check(fn(*args), name)
    e1.record()
    _timer.records.append((name, flops, nbytes, e0, e1, alg_flops))


def _p(t):
    return None if t is None else t.data_ptr()


def _dev_f32(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor — the DVG hot path has no CPU fallback")
    if t.dtype != torch.float32:
        raise RuntimeError(f"{name}: expected float32, got {t.dtype}")


def nhwc_empty(n: int, c: int, h: int, w: int, device) -> torch.Tensor:
    """(N,C,H,W)-shaped view of a fresh NHWC buffer."""
    return torch.empty((n, h, w, c), device=device, dtype=torch.float32).permute(0, 3, 1, 2)


def is_nhwc(t: torch.Tensor) -> bool:
    if t.dim() != 4:
        return False
    n, c, h, w = t.shape
    return t.stride() == (h * w * c, 1, w * c, c) or (c == 1 and t.is_contiguous()) or \
        (h == 1 and w == 1 and t.is_contiguous())


def to_nhwc(t: torch.Tensor) -> torch.Tensor:
    """Return `t` (N,C,H,W) as an NHWC-in-memory tensor, converting with the layout kernel if needed."""
    _dev_f32(t, "to_nhwc")
    if is_nhwc(t):
        return t
    src = t if t.is_contiguous() else t.contiguous()
    n, c, h, w = src.shape
    out = nhwc_empty(n, c, h, w, t.device)
    check(lib().dvg_nchw_to_nhwc(_p(src), _p(out), n, c, h, w, _stream()), "nchw_to_nhwc")
    return out


def to_nchw(t: torch.Tensor) -> torch.Tensor:
    """Contiguous NCHW copy of an NHWC-in-memory tensor."""
    _dev_f32(t, "to_nchw")
    if t.is_contiguous():
        return t
    if not is_nhwc(t):
        return t.contiguous()
    n, c, h, w = t.shape
    out = torch.empty((n, c, h, w), device=t.device, dtype=torch.float32)
    check(lib().dvg_nhwc_to_nchw(_p(t), _p(out), n, c, h, w, _stream()), "nhwc_to_nchw")
    return out


class tile_policy:
    """`with ops.tile_policy(energy=True):` - the launches (and hipGraph captures) inside pick the energy-lean tiles
    (dvg_set_tile_policy: the 128 x 128 batched-GEMM tile, 8 x 16 pixel tiles from one workgroup per CU on) that pay when several
    independent chains keep the board at its power cap; the default (latency) tiles are what makes one chain fastest.  Results
    under the two policies agree to fp32 rounding (the tile shape fixes how a K sum is cut), not bit for bit; compare like with like.
    rollout.ConcurrentRollouts / GraphedSampler capture their chains under it when more than one is in flight."""

    def __init__(self, energy: bool = True):
        self.energy = bool(energy)

    def __enter__(self):
        self.prev = lib().dvg_tile_policy()
        lib().dvg_set_tile_policy(int(self.energy))
        return self

    def __exit__(self, *exc):
        lib().dvg_set_tile_policy(self.prev)
