"""Fused multi-tensor Adam for the four optimisers of train.py:95-106.

`FusedAdam(params, lr=...)` is a `torch.optim.Optimizer` with torch.optim.Adam's arithmetic (non-amsgrad) and
state layout (`step`, `exp_avg`, `exp_avg_sq` per parameter, so `state_dict()` is interchangeable with
torch.optim.Adam's and `MultiStepLR` drives `param_groups[...]['lr']` as in train.py:105-106), but every
parameter group lives in ONE flat fp32 buffer: parameters are re-pointed at views of it, the moments are views of
two more, the GRADIENTS are views of a fourth (`p.grad` is set to its view, so backward accumulates straight into
the flat buffer) and a step is ONE `dvg_adam_step` launch per group.

`FlatArena`: several optimisers can carve their groups out of one shared arena (train.Trainer does: GP, likelihood,
LSTM, decoder, encoder in that order), so that the data-parallel gradient all-reduce (dvg_amd/parallel.py) runs on
contiguous ranges of ONE buffer - no gather into a communication buffer, no scatter back, no copy into the
optimiser's buffer.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from . import ops
from ._lib import check, lib


class FlatArena:
    """Four flat fp32 buffers (param, grad, exp_avg, exp_avg_sq) of `numel` floats; `alloc` hands out ranges."""

    def __init__(self, numel: int, device):
        self.p = torch.zeros(numel, device=device)
        self.g = torch.zeros(numel, device=device)
        self.m = torch.zeros(numel, device=device)
        self.v = torch.zeros(numel, device=device)
        self.used = 0

    @staticmethod
    def padded(n: int) -> int:
        return (n + 3) // 4 * 4          # every view 16-byte aligned

    @classmethod
    def size_for(cls, params) -> int:
        return sum(cls.padded(p.numel()) for p in params if p.requires_grad)

    def alloc(self, n: int) -> int:
        if self.used + n > self.p.numel():
            raise RuntimeError("FlatArena: out of space")
        off, self.used = self.used, self.used + n
        return off


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, arena: Optional[FlatArena] = None):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("FusedAdam: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.arena = arena
        self._flat = {}   # group index -> dict(p, g, m, v, params, offsets, gviews, tdev, lo, hi)
        self._captured_groups = []   # groups stepped while a hipGraph was being captured (see after_graph_replay)
        if all(p.is_cuda for g in self.param_groups for p in g["params"]):
            for gi, group in enumerate(self.param_groups):   # flat storage right away: p.grad are views from the start
                self._build(gi, group)

    # ---- flat storage -------------------------------------------------------------------------------
    def _build(self, gi: int, group) -> dict:
        params: List[torch.nn.Parameter] = [p for p in group["params"] if p.requires_grad]
        if not params:
            return {}
        dev = params[0].device
        for p in params:
            if p.dtype != torch.float32 or p.device != dev:
                raise RuntimeError("FusedAdam: parameters of a group must be fp32 on one device")
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += FlatArena.padded(p.numel())
        if self.arena is not None:
            lo = self.arena.alloc(n)
            a = self.arena
            flat_p, flat_g, flat_m, flat_v = (t[lo:lo + n] for t in (a.p, a.g, a.m, a.v))
        else:
            lo = 0
            flat_p = torch.zeros(n, device=dev)
            flat_m, flat_v, flat_g = torch.zeros_like(flat_p), torch.zeros_like(flat_p), torch.zeros_like(flat_p)
        steps = {}
        with torch.no_grad():
            for p, o in zip(params, offs):
                view = flat_p[o:o + p.numel()].view(p.shape)
                view.copy_(p)
                st = self.state.get(p, {})
                if "exp_avg" in st:                 # state restored by load_state_dict
                    flat_m[o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                    flat_v[o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                    steps[p] = float(st["step"])
                p.data = view                       # the module keeps the same Parameter object
        f = dict(p=flat_p, g=flat_g, m=flat_m, v=flat_v, params=params, offsets=offs, lo=lo, hi=lo + n,
                 tdev=torch.zeros(1, dtype=torch.int32, device=dev),     # device-side step count of the group
                 gviews=[flat_g[o:o + p.numel()].view(p.shape) for p, o in zip(params, offs)])
        for p, o, gv in zip(params, offs, f["gviews"]):
            self.state[p] = {"step": torch.tensor(steps.get(p, 0.0)), "exp_avg": flat_m[o:o + p.numel()].view(p.shape),
                             "exp_avg_sq": flat_v[o:o + p.numel()].view(p.shape)}
            if p.grad is not None:
                gv.copy_(p.grad)
            p.grad = gv                             # backward accumulates into the flat gradient buffer
        self._flat[gi] = f
        return f

    def flat_range(self, gi: int = 0):
        """(lo, hi) of group `gi` in the arena (or in its own flat buffer)."""
        f = self._flat[gi]
        return f["lo"], f["hi"]

    def flat_grad(self, gi: int = 0) -> torch.Tensor:
        return self._flat[gi]["g"]

    def state_dict(self):
        """torch.optim layout, but every per-parameter tensor is a COPY that owns its storage: `exp_avg` / `exp_avg_sq` are
        views of the flat moment buffers (of the shared arena under train.Trainer), and torch.save writes the whole storage
        behind a view - the GP optimiser's entry of a checkpoint (train.py:385) would carry the Adam moments of every
        module (+2 x all parameter bytes)."""
        sd = super().state_dict()
        sd["state"] = {k: {n: (t.detach().clone() if torch.is_tensor(t) else t) for n, t in st.items()}
                       for k, st in sd["state"].items()}
        return sd

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        # torch aliases (does not copy) the tensors of `state_dict`: copy them into our flat buffers right away
        if self.arena is not None and self._flat:
            for gi, f in self._flat.items():     # arena ranges are already ours: refresh moments / steps in place
                for p, o in zip(f["params"], f["offsets"]):
                    st = self.state[p]
                    f["m"][o:o + p.numel()].copy_(st["exp_avg"].reshape(-1))
                    f["v"][o:o + p.numel()].copy_(st["exp_avg_sq"].reshape(-1))
                    self.state[p] = {"step": torch.tensor(float(st["step"])),
                                     "exp_avg": f["m"][o:o + p.numel()].view(p.shape),
                                     "exp_avg_sq": f["v"][o:o + p.numel()].view(p.shape)}
            return
        self._flat.clear()
        for gi, group in enumerate(self.param_groups):
            self._build(gi, group)

    @torch.no_grad()
    def zero_grad(self, set_to_none: bool = False):
        """One fill per group; `p.grad` stays (or becomes again) the view of the flat gradient buffer.  Unlike
        torch.optim the default is set_to_none=False: a None gradient makes autograd allocate a fresh tensor per parameter
        and step() copy it back into the flat buffer (still correct - and it keeps torch.optim.Adam's "a parameter without
        a gradient is skipped" semantics, which is why set_to_none=True is honoured when asked for)."""
        for gi, group in enumerate(self.param_groups):
            f = self._flat.get(gi)
            if not f or set_to_none:
                for p in group["params"]:
                    p.grad = None
                continue
            f["g"].zero_()
            for p, gv in zip(f["params"], f["gviews"]):
                if p.grad is not gv:
                    p.grad = gv

    def _restore_grad_views(self, f) -> None:
        for p, gv in zip(f["params"], f["gviews"]):
            if p.grad is not gv:
                p.grad = gv

    # ---- hipGraph support -------------------------------------------------------------------------------
    def begin_capture(self) -> None:
        """Call right before capturing a graph that contains step(): the device step counts are synchronised with the
        host ones (the captured kernels then increment and read them on the device).
        INVARIANT: the device counts `tdev` are valid only from here through the replays of the graph captured
        next.  An EAGER step() in between advances the host count only (it passes the count as an argument), so a replay
        after it would apply stale bias corrections: step() marks the captured graph stale and after_graph_replay() raises.
        Re-capture (begin_capture again) to continue with graphs after eager steps."""
        self._captured_groups = []
        self._graph_stale = False
        for gi, f in self._flat.items():
            if f:
                f.pop("ticked", None)
                f["tdev"].fill_(int(self.state[f["params"][0]]["step"]))

    def end_capture(self) -> None:
        """Call after the capture: every group whose device step count a captured zero_grads() launch advanced (`ticked`)
        must also have been stepped inside the same capture - otherwise each replay would advance the count without a step
        and later bias corrections would be wrong without any error."""
        for gi, f in self._flat.items():
            if f and f.pop("ticked", False):
                raise RuntimeError(f"FusedAdam: group {gi} had its device step count advanced by a captured zero_grads() "
                                   "but was not stepped in the same capture")

    def check_graph_fresh(self) -> None:
        """Call BEFORE replaying a captured graph: raises when an eager step() ran since the capture (nothing has executed yet)."""
        if getattr(self, "_graph_stale", False) and self._captured_groups:
            raise RuntimeError("FusedAdam: an eager step() ran since the graph was captured: its device-side step counts are "
                               "stale (a replay would apply wrong Adam bias corrections).  Re-capture after eager steps.")

    def after_graph_replay(self) -> None:
        """The replayed kernels stepped the parameters through raw pointers: advance the host-side step counts (state_dict,
        torch.optim compatibility) and the version counters the packed-weight caches key on.  Raises when an eager step()
        ran since the capture (see begin_capture: the replay just applied stale bias corrections)."""
        if getattr(self, "_graph_stale", False) and self._captured_groups:
            raise RuntimeError("FusedAdam: a captured graph was replayed after an eager step(): its device-side step counts "
                               "are stale (wrong Adam bias corrections were applied).  Re-capture after eager steps.")
        for gi in self._captured_groups:
            for p in self._flat[gi]["params"]:
                torch.autograd.graph.increment_version(p)
                self.state[p]["step"] += 1

    # ---- step ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            f = self._flat.get(gi)
            if f is None or [id(p) for p in f.get("params", [])] != [id(p) for p in group["params"] if p.requires_grad]:
                if self.arena is not None and f is not None:
                    raise RuntimeError("FusedAdam: the parameters of an arena-backed group cannot change")
                f = self._build(gi, group)
            if not f:
                continue
            have = [p.grad is not None for p in f["params"]]
            if not any(have):
                continue
            b1, b2 = group["betas"]
            hyper = (float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]))
            steps = {int(self.state[p]["step"]) for p in f["params"]}
            capturing = torch.cuda.is_current_stream_capturing()
            if all(have) and len(steps) == 1:
                # the whole group in one launch; the step count lives on the device so that a captured hipGraph
                # (train.GraphedIteration) applies fresh bias corrections at every replay.  Gradients normally ARE the
                # views of the flat buffer; a gradient that was re-created (module.zero_grad() sets None) is copied in.
                stray = [(gv, p.grad) for p, gv in zip(f["params"], f["gviews"])
                         if p.grad is not gv and p.grad.data_ptr() != gv.data_ptr()]
                if stray:
                    torch._foreach_copy_([s[0] for s in stray], [s[1] for s in stray])
                    for p, gv in zip(f["params"], f["gviews"]):
                        p.grad = gv
                t = steps.pop() + 1
                if capturing:
                    # the count lives on the device (begin_capture synchronised it): advanced by the zero_grads() launch of
                    # this capture when there was one since the group's last step, else by a one-element add
                    if not f.pop("ticked", False):
                        f["tdev"].add_(1)
                    tdev = f["tdev"]
                else:
                    tdev = None                      # eager: the host count goes in as an argument, no device-side bookkeeping
                    f.pop("ticked", None)
                    if gi in self._captured_groups:  # a captured graph holds this group's device count: now behind the host's
                        self._graph_stale = True
                check(lib().dvg_adam_step(ops._p(f["p"]), ops._p(f["g"]), ops._p(f["m"]), ops._p(f["v"]),
                                          f["p"].numel(), *hyper, t, ops._p(tdev), ops._stream()), "dvg_adam_step")
                touched = f["params"]
                if capturing:
                    self._captured_groups.append(gi)
            else:
                # torch.optim.Adam semantics for a partially used group: parameters without a gradient are skipped
                # (no moment decay, no step count) - one launch per parameter that has one
                if capturing:
                    raise RuntimeError("FusedAdam: a partially used parameter group cannot be captured in a hipGraph")
                touched = [p for p in f["params"] if p.grad is not None]
                for p, gv in zip(f["params"], f["gviews"]):
                    if p.grad is None:
                        continue
                    st = self.state[p]
                    if p.grad is not gv and p.grad.data_ptr() != gv.data_ptr():
                        gv.copy_(p.grad)
                    check(lib().dvg_adam_step(ops._p(p), ops._p(gv), ops._p(st["exp_avg"]), ops._p(st["exp_avg_sq"]),
                                              p.numel(), *hyper, int(st["step"]) + 1, None, ops._stream()),
                          "dvg_adam_step")
            for p in touched:
                # the kernel wrote through raw pointers; also while capturing, so that code captured AFTER this step
                # re-packs its weights instead of reusing the packs from before the step
                torch.autograd.graph.increment_version(p)
                if not capturing:                # a capture executes nothing: after_graph_replay() counts the replays
                    self.state[p]["step"] += 1
        return loss


@torch.no_grad()
def zero_grads(optimizers) -> None:
    """`zero_grad()` of several FusedAdam optimisers whose groups live in ONE arena (train.py:201-203: encoder, decoder and
    frame predictor at the top of train_model): adjacent ranges of the gradient arena are zeroed by one launch
    (dvg_zero_tick), and during a hipGraph capture the same launch advances the device-side step counts of those groups
    (step() then skips its one-element increment).  Optimisers without flat arena storage fall back to their own zero_grad."""
    spans = []
    for o in optimizers:
        fl = getattr(o, "_flat", None)
        if not isinstance(o, FusedAdam) or o.arena is None or not fl or any(not f for f in fl.values()):
            o.zero_grad()
            continue
        for f in fl.values():
            spans.append((o.arena, f["lo"], f["hi"], f, o))
    capturing = torch.cuda.is_current_stream_capturing()
    spans.sort(key=lambda s_: (id(s_[0]), s_[1]))
    i = 0
    while i < len(spans):
        arena, lo, hi = spans[i][0], spans[i][1], spans[i][2]
        group = [spans[i]]
        while (i + 1 < len(spans) and spans[i + 1][0] is arena and spans[i + 1][1] == hi and len(group) < 4):
            i += 1
            hi = spans[i][2]
            group.append(spans[i])
        ticks = [ops._p(s_[3]["tdev"]) if capturing else None for s_ in group] + [None] * (4 - len(group))
        check(lib().dvg_zero_tick(ops._p(arena.g[lo:hi]), hi - lo, *ticks, ops._stream()), "dvg_zero_tick")
        for s_ in group:
            if capturing:
                s_[3]["ticked"] = True
            s_[4]._restore_grad_views(s_[3])
        i += 1
