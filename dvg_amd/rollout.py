"""The inference rollout of generate_frames.py (the path the headline metric is quoted on).

`sample_rollout` is ONE pass of the `for s in range(nsample)` body of make_gifs
(generate_frames.py:143-177) with all five modules in eval mode:

    for i in 1..n_eval-1:
        h, skip = encoder(x_in)                 # skip kept only while i < n_past (or last_frame_skip)
        i <  n_past: step the LSTM on h, discard its output, x_in = x[i]          (:159-164)
        i >= n_past: h_pred = lstm(h)
                     i % 15 == 0: x_in = decoder([likelihood(gp(h^T)).rsample()^T, skip])   (:168-171)
                     else:        x_in = decoder([h_pred, skip])                            (:174)

The reference also runs `encoder(x[i])` on conditioning steps and discards it (:160-161); in eval
mode that call has no side effect, so it is skipped here.
Index bookkeeping (row K of SURVEY.md §8) is integer logic kept in `trigger_steps` / the skip rule.
"""
from __future__ import annotations

import os

from typing import Dict, List, Optional, Sequence

import torch

from . import fused, ops


def drop_version_keyed_caches() -> None:
    """Forget every tensor the module-level caches hold (packed / transposed / Winograd-domain weights, BatchNorm folds,
    embed-folded cell weights, GEMM-operand forms, hoisted skip halves and projections, device index maps, the LSTM's shared
    zero state).  For the one situation in which an entry can be WRONG although its key matches: a hipGraph capture that
    raised part-way.  During a capture a cache miss allocates from the graph's private pool and records the kernels that
    would fill the tensor - nothing executes - and the entry is stored under the current parameter version; when the
    capture is abandoned the pool is freed, but the entry would still be hit by the next eager call, which would read
    never-written (and freed) memory.  Everything here is rebuilt on demand from the parameters.
    The deferred weight-gradient queues go with them: a capture that dies INSIDE backward() skips autograd's
    end-of-backward callback, so operands queued during it - pointers into the freed pool, never written - would be
    flushed into .grad by the next eager backward (or block its flush: the `flush queued` flag stays set)."""
    from . import autograd as ag
    from .models import lstm as lstm_mod
    ag.drop_deferred_wgrads()
    ag._pack_cache.clear()
    for slot in list(fused._cache.values()):
        slot.clear()
    ops._WMAT_CACHE.clear()
    ops._MAP_CACHE.clear()
    ops._SPLITK_WS.clear()
    ops.clear_skip_proj_cache()
    fused.clear_skip_hoist_cache()
    lstm_mod._FOLD_CACHE.clear()
    _ZERO_STATE.clear()


# make_gifs: the prediction steps before the first GP trigger step once per batch instead of once per sample (GraphedSampler)
SHARE_PREFIX = True      # module attribute (tests toggle it; generate_frames.py --no_share_prefix); an environment switch until r06


def trigger_steps(n_past: int, n_eval: int, period: int = 15) -> List[int]:
    return [i for i in range(n_past, n_eval) if i % period == 0]


def _adjacent_view(frames):
    """If the frames are consecutive contiguous slices of ONE buffer (data.batch_device(), GraphedRollout's static inputs),
    the batch of all of them is a view: no concat launch per rollout.  None otherwise."""
    f0 = frames[0]
    if not all(t.is_contiguous() and t.shape == f0.shape and t.dtype == f0.dtype for t in frames):
        return None
    step = f0.numel() * f0.element_size()
    st = f0.untyped_storage()
    for i, t in enumerate(frames):
        if t.untyped_storage().data_ptr() != st.data_ptr() or t.data_ptr() != f0.data_ptr() + i * step:
            return None
    shape = (len(frames) * f0.shape[0],) + tuple(f0.shape[1:])
    return torch.as_strided(f0, shape, torch.empty(shape, device="meta").stride(), f0.storage_offset())


_ZERO_STATE = {}

# Graph captures use the thread-local error mode: what OTHER threads do while this thread captures (torch.distributed's
# watchdog polling the events of earlier collectives when a process group exists) is none of the capture's business.
CAPTURE_KW = {"capture_error_mode": "thread_local"}


def snapshot_eager_caches() -> list:
    """Strong references to every tensor the module-level caches hold right now: the LSTM's shared zero state, packed /
    transposed / Winograd-domain weights, BatchNorm folds, the embed-folded first-cell weights, GEMM-operand forms of the
    dense ends.  A captured hipGraph reads such tensors by RAW POINTER when they were created eagerly during its warm-up;
    the caches evict (`.clear()` past 16 / 64 / 4096 entries) and replace entries when a parameter version changes
    (load_state_dict, an optimiser step), after which the allocator may hand the block to someone else while live graphs
    still read it.  Every graph holder keeps the snapshot taken right after its capture for as long as it lives."""
    from . import autograd as ag
    from .models import lstm as lstm_mod
    keep = []

    def walk(o):
        if torch.is_tensor(o):
            keep.append(o)
        elif isinstance(o, (tuple, list)):
            for v in o:
                walk(v)
        elif isinstance(o, dict):
            for v in o.values():
                walk(v)
        elif hasattr(o, "__dict__") and type(o).__module__.startswith("dvg_amd"):
            walk(vars(o))          # ops.WinoV and similar small holders

    walk(list(_ZERO_STATE.values()))
    walk(list(ops._WMAT_CACHE.values()))
    walk(list(ops._SKIP_PROJ_CACHE.values()))
    walk(list(lstm_mod._FOLD_CACHE.values()))
    walk(list(ag._pack_cache.values()))
    walk([dict(d) for d in list(fused._cache.values())])
    if not torch.cuda.is_current_stream_capturing():
        ops.evict_captured_workspaces()      # the capture that just ended no longer needs its split-K workspace OBJECTS
    return keep


def _zero_hidden(frame_predictor):
    """init_hidden() (lstm.py:58-63) without its 2 x n_layers fill launches per rollout: one cached all-zero tensor per
    (device, batch, hidden size).  Safe to share: no kernel of the recurrent path writes its state in place (dvg_lstm_cell
    rejects in-place updates), and the tensors are only ever replaced in the `hidden` list."""
    if not hasattr(frame_predictor, "step_state_only"):
        return frame_predictor.init_hidden()
    dev = frame_predictor.embed.weight.device
    key = (dev, frame_predictor.batch_size, frame_predictor.hidden_size)
    z = _ZERO_STATE.get(key)
    if z is None:
        if torch.cuda.is_current_stream_capturing():
            return frame_predictor.init_hidden()    # never cache a tensor that lives in a graph's private pool
        if len(_ZERO_STATE) > 16:
            _ZERO_STATE.clear()
        z = _ZERO_STATE[key] = torch.zeros(frame_predictor.batch_size, frame_predictor.hidden_size, device=dev)
    return [(z, z) for _ in range(frame_predictor.n_layers)]


# Skip tensors the rollout never reads are not stored (encoder.encode(x, skips_from), VggEncoder.features): of the conditioning
# batch only the LAST frame's skips survive, and once the skip is frozen (generate_frames.py:154-157) the encoder's skips of
# every predicted frame are discarded by the caller (`h, _ = h`).  ELIDE_SKIPS = False: every call stores all of them (module attribute; an environment switch until r06).
ELIDE_SKIPS = True


def _encode(encoder, x, skips_from=0):
    """encoder(x) -> (h, skips); with skips_from = k > 0 the skips hold the images [k, N) only (None when k == N)."""
    if skips_from and ELIDE_SKIPS and hasattr(encoder, "encode") and not encoder.training and not torch.is_grad_enabled():
        return encoder.encode(x, skips_from)
    h, skips = encoder(x)
    if skips_from:
        skips = [s[skips_from:] if skips_from < s.shape[0] else None for s in skips]
    return h, skips


def _encode_conditioning(encoder, x, n_past, last_frame_skip):
    """Eval-mode only: BatchNorm uses running statistics, so encoder outputs are independent across samples and the
    n_past-1 conditioning frames x[0..n_past-2] (all known before the rollout starts) can go through the encoder as
    ONE batch of B*(n_past-1) frames — bit-identical per-sample math, 9 passes' worth of launches folded into one and
    9x more tiles per launch for the deep 8x8 layers.  Returns ([h_1 .. h_{n_past-1}], skip of the last step)."""
    b = x[0].shape[0]
    frames = _adjacent_view(x[:n_past - 1])
    if frames is None:
        frames = torch.cat([x[i] for i in range(n_past - 1)], 0)
    last = n_past - 2
    h_all, skip = _encode(encoder, frames, last * b)      # the skips of the last frame x[n_past-2] alone (b images each)
    hs = [h_all[i * b:(i + 1) * b] for i in range(n_past - 1)]
    return hs, skip


# Streams are made ONCE per process and shared by every graph holder of this module.  torch hands out streams from a pool of 32
# per device, round-robin and without telling: a process that builds many samplers (bench.py: two families x (three chains +
# make_gifs) + C1) wraps around the pool, after which two "concurrent" chains may sit on ONE hip stream (silently serial), or a
# chain's stream may be the one torch captures graphs on - the r06 bench died with a segmentation fault inside
# hipGraphLaunch (hip::Graph::UpdateStreams) at exactly that point.  Work on one stream runs in order, so sharing the streams
# between holders is safe; it only orders work that a caller issues from different holders at the same time.
_streams = {}


def pooled_stream(role: str, index: int = 0) -> "torch.cuda.Stream":
    key = (torch.cuda.current_device(), role, index)
    if key not in _streams:
        _streams[key] = torch.cuda.Stream()
    return _streams[key]


def _hoist_stream():
    return pooled_stream("hoist")


def _step_discard(frame_predictor, h):
    """LSTM stepped on a conditioning frame, output discarded (generate_frames.py:125,162): skip the output GEMM."""
    if hasattr(frame_predictor, "step_state_only"):
        frame_predictor.step_state_only(h)
    else:
        frame_predictor(h)


@torch.no_grad()
def condition(encoder, frame_predictor, x: Sequence[torch.Tensor], n_past: int, last_frame_skip: bool = False,
              batch_conditioning: bool = True, decoder=None) -> dict:
    """The part of a sample rollout that does not depend on the sample (generate_frames.py:147-162 for i < n_past):
    the LSTM is reset and stepped on the encodings of the conditioning frames x[0..n_past-2] (outputs discarded), the
    skip tensors are those of x[n_past-2].  Deterministic in eval mode, so `make_gifs` (nsample rollouts of the SAME
    batch) computes it once per batch and draws every sample with `sample_from`."""
    frame_predictor.hidden = _zero_hidden(frame_predictor)
    skip = None
    if batch_conditioning and n_past >= 3 and not encoder.training:
        hs, skip = _encode_conditioning(encoder, x, n_past, last_frame_skip)
        side = None
        if (decoder is not None and not last_frame_skip and not decoder.training and fused.SKIP_HOIST and
                hasattr(decoder, "precompute_frozen_skips")):
            # The skip tensors are final here (generate_frames.py:154-157) and the LSTM warm-up below is a serial chain of
            # small latency-bound launches: the decoder's loop-invariant skip halves (fused._hoisted_skip) are computed
            # meanwhile on a second stream (a parallel branch of the hipGraph when captured) instead of on the critical
            # path of the first decoder call.
            side, cur = _hoist_stream(), torch.cuda.current_stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                fused.declare_frozen_skips(skip)
                decoder.precompute_frozen_skips(skip)
        for i in range(1, n_past):
            _step_discard(frame_predictor, hs[i - 1])   # LSTM stepped on conditioning frames, output discarded (:162)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
    else:
        for i in range(1, n_past):
            h, skip = encoder(x[i - 1])
            _step_discard(frame_predictor, h)
    return {"hidden": list(frame_predictor.hidden), "skip": skip, "frames": [x[i] for i in range(n_past)]}


@torch.no_grad()
def sample_from(state: dict, encoder, decoder, frame_predictor, gp_layer, likelihood, n_past: int, n_eval: int,
                last_frame_skip: bool = False, period: int = 15,
                eps_by_step: Optional[Dict[int, torch.Tensor]] = None) -> List[torch.Tensor]:
    """The prediction phase of one sample (generate_frames.py:163-176), starting from a `condition()` state."""
    frame_predictor.hidden = list(state["hidden"])   # lstm_cell returns new tensors: the saved state is never mutated
    frames = list(state["frames"])
    skip = state["skip"]
    x_in = frames[n_past - 1]
    for i in range(n_past, n_eval):
        keep = last_frame_skip or skip is None
        h, sk = _encode(encoder, x_in, 0 if keep else x_in.shape[0])   # frozen skip: this frame's skips are discarded (:157)
        if keep:
            skip = sk
        if i == n_past and not last_frame_skip and not decoder.training:
            fused.declare_frozen_skips(skip)    # frozen from here on: decoder blocks hoist their skip halves now
        h_pred = frame_predictor(h)
        if period and i % period == 0:
            pred = likelihood(gp_layer(h.transpose(0, 1).view(gp_layer.num_dims, h.shape[0], 1)))
            z = pred.rsample(None if eps_by_step is None else eps_by_step[i])
            x_in = decoder([z.transpose(0, 1), skip])
        else:
            x_in = decoder([h_pred, skip])
        frames.append(x_in)
    return frames


@torch.no_grad()
def sample_rollout(encoder, decoder, frame_predictor, gp_layer, likelihood, x: Sequence[torch.Tensor], n_past: int,
                   n_eval: int, last_frame_skip: bool = False, period: int = 15,
                   eps_by_step: Optional[Dict[int, torch.Tensor]] = None,
                   batch_conditioning: bool = True) -> List[torch.Tensor]:
    """Returns the n_eval frames [x0, ..] of one sample (conditioning frames are the inputs themselves): one complete
    rollout = condition() + sample_from()."""
    state = condition(encoder, frame_predictor, x, n_past, last_frame_skip, batch_conditioning, decoder)
    return sample_from(state, encoder, decoder, frame_predictor, gp_layer, likelihood, n_past, n_eval, last_frame_skip,
                       period, eps_by_step)


@torch.no_grad()
def posterior_rollout(encoder, decoder, frame_predictor, gp_layer, likelihood, x, n_past, n_eval,
                      last_frame_skip=False) -> List[torch.Tensor]:
    """generate_frames.py:110-134: the GP is fed the LSTM output and its predictive MEAN is decoded."""
    frame_predictor.hidden = frame_predictor.init_hidden()
    frames = [x[0]]
    x_in = x[0]
    skip = None
    for i in range(1, n_eval):
        keep = last_frame_skip or i < n_past
        h, sk = _encode(encoder, x_in, 0 if keep else x_in.shape[0])
        if keep:
            skip = sk
        if i == n_past and not last_frame_skip and not decoder.training:
            fused.declare_frozen_skips(skip)
        if i < n_past:
            _step_discard(frame_predictor, h)
            x_in = x[i]
        else:
            h_pred = frame_predictor(h)
            pred = likelihood(gp_layer(h_pred.transpose(0, 1).view(gp_layer.num_dims, h_pred.shape[0], 1)))
            x_in = decoder([pred.mean.transpose(0, 1), skip])
        frames.append(x_in)
    return frames


@torch.no_grad()
def posterior_from(state: dict, encoder, decoder, frame_predictor, gp_layer, likelihood, n_past: int, n_eval: int,
                   last_frame_skip: bool = False) -> List[torch.Tensor]:
    """`posterior_rollout`'s prediction steps n_past ... n_eval - 1 from a `condition()` state.  Its conditioning phase
    (generate_frames.py:113-121 for i < n_past: the encoder on x[i-1], the LSTM stepped with the output discarded, the skip
    tensors of the last such frame, x[i] handed on) is the same computation as the samples' (:147-162), so `make_gifs` runs it
    once for both."""
    frame_predictor.hidden = list(state["hidden"])
    frames = list(state["frames"])
    skip = state["skip"]
    x_in = frames[n_past - 1]
    for i in range(n_past, n_eval):
        keep = last_frame_skip or skip is None
        h, sk = _encode(encoder, x_in, 0 if keep else x_in.shape[0])
        if keep:
            skip = sk
        if i == n_past and not last_frame_skip and not decoder.training:
            fused.declare_frozen_skips(skip)
        h_pred = frame_predictor(h)
        pred = likelihood(gp_layer(h_pred.transpose(0, 1).view(gp_layer.num_dims, h_pred.shape[0], 1)))
        x_in = decoder([pred.mean.transpose(0, 1), skip])
        frames.append(x_in)
    return frames


# ---- GPtrigger_gen (generate_frames.py:220-298) without a host round trip per step --------------------------------------------
# The reference pulls the predictive variance to the host at every step, slides the 12-long window in numpy and branches in
# Python; it re-runs the 12 warm-up steps for every batch index although only the COLUMN of the variance norms they record
# depends on the index (:275), and it encodes x_in two or three times per step (`var_value`, then `generation` or the trigger
# branch: the same eval-mode encoder on the same frame).  Here: the warm-up once per batch with the norms of every sample
# (`trigger_warmup`), one encoder call per step, the decision and the branch select on the device (ops.gp_trigger_step /
# gp_trigger_select: both branches' cheap parts - the GP sample, the LSTM step - are computed, the flag picks the latent to
# decode and the recurrent state that survives), so that the main loop is capturable (`GraphedTrigger`) and the value /
# threshold / trigger logs are read back once per index.
@torch.no_grad()
def trigger_warmup(encoder, decoder, frame_predictor, gp_layer, likelihood, x0, warmup: int = 12, skip_steps: int = 5) -> dict:
    """Loop steps 0 .. warmup-1 of GPtrigger_gen (:266-280) for the whole batch: autoregressive from x0, skip tensors of the steps
    i < skip_steps (:268-269), every step `generation` (:220-224).  norms[i, b] = the value the reference records at step i
    for batch index b (:275)."""
    b = x0.shape[0]
    frame_predictor.hidden = _zero_hidden(frame_predictor)
    x_in, skip, norms, frames = x0, None, [], []
    for i in range(warmup):
        keep = i < skip_steps
        h, sk = _encode(encoder, x_in, 0 if keep else b)
        if keep:
            skip = sk
        if i == skip_steps - 1 and not decoder.training:
            fused.declare_frozen_skips(skip)            # final from here on (:268-271)
        pred = likelihood(gp_layer(h.transpose(0, 1).view(gp_layer.num_dims, b, 1)))
        norms.append(ops.gp_var_norms(pred.variance))
        x_in = decoder([frame_predictor(h), skip])      # generation(): its encoder call is the one above (eval mode)
        frames.append(x_in)
    return {"x_in": x_in, "skip": skip, "hidden": list(frame_predictor.hidden), "norms": torch.stack(norms), "frames": frames}


@torch.no_grad()
def trigger_body(state: dict, encoder, decoder, frame_predictor, gp_layer, likelihood, ctx, coef: float, eps_all, log: dict,
                 warmup: int = 12, total: int = 105, probe: int = 3) -> List[torch.Tensor]:
    """Loop steps warmup .. total-1 (:285-297) from a `trigger_warmup` state.  ctx: the window (float32, device), initialised
    by the caller to norms[:, index] and slid in place; eps_all[i - warmup]: the base sample (D,B) of step i; log: device
    tensors `flag` (1 int32), `values` / `thresholds` (total float32), `flags` (total int32).  Nothing here syncs."""
    b = state["x_in"].shape[0]
    if b <= probe:
        raise IndexError(f"GPtrigger_gen reads sample [{probe}] of the batch (generate_frames.py:230): batch_size must be > {probe}")
    frame_predictor.hidden = list(state["hidden"])
    x_in, skip, frames = state["x_in"], state["skip"], []
    for i in range(warmup, total):
        h, _ = _encode(encoder, x_in, b)
        pred = likelihood(gp_layer(h.transpose(0, 1).view(gp_layer.num_dims, b, 1)))
        z = pred.rsample(eps_all[i - warmup])                       # (D,B); the same launch leaves the variance
        ops.gp_trigger_step(pred.variance, probe, ctx, coef, log["flag"], log["values"], log["thresholds"], log["flags"], i)
        old = [t for hc in frame_predictor.hidden for t in hc]
        h_pred = frame_predictor(h)
        new = [t for hc in frame_predictor.hidden for t in hc]
        vec, st = ops.gp_trigger_select(log["flag"], z, h_pred, old, new)
        frame_predictor.hidden = [(st[2 * l], st[2 * l + 1]) for l in range(len(st) // 2)]
        x_in = decoder([vec, skip])
        frames.append(x_in)
    return frames


def trigger_log(total: int, device) -> dict:
    return {"flag": torch.zeros(1, dtype=torch.int32, device=device), "values": torch.zeros(total, device=device),
            "thresholds": torch.zeros(total, device=device), "flags": torch.zeros(total, dtype=torch.int32, device=device)}


class GraphedTrigger:
    """GPtrigger_gen for one batch as two hipGraphs: the warm-up (once per batch) and the main loop (once per batch index; it
    reads the warm-up graph's state - frames, LSTM state, skip tensors and the decoder's hoisted skip halves - and its own
    static window / base samples / logs).  `warm(x0)` then `run(index)`; one device-to-host copy of the logs per index."""

    def __init__(self, encoder, decoder, frame_predictor, gp_layer, likelihood, x0, warmup=12, total=105, depth=1,
                 skip_steps=5, probe=3):
        self._mods = (encoder, decoder, frame_predictor, gp_layer, likelihood)
        self.warmup, self.total, self.probe, self.skip_steps = warmup, total, probe, skip_steps
        self.coef = 2 + 0.01 * depth
        dev, b, d = x0.device, x0.shape[0], gp_layer.num_dims
        self.x0 = x0.contiguous().clone()
        self.eps = torch.zeros(max(1, total - warmup), d, b, device=dev)
        self.ctx = torch.zeros(warmup, device=dev)
        self.log = trigger_log(total, dev)
        side = pooled_stream("warmup")
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # first-launch work (weight packs, LDS attributes, BN folds) must not be captured
            st = self._warm()
            self._body(st)
        torch.cuda.current_stream().wait_stream(side)
        ops.clear_skip_proj_cache()
        fused.clear_skip_hoist_cache()
        self.warm_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.warm_graph, **CAPTURE_KW):
            self.state = self._warm()
            self.warm_stack = torch.stack(self.state["frames"])
        self._keep_w = snapshot_eager_caches()
        # (the skip-dependent cache entries the warm-up graph created stay while the body is captured: it reads those buffers)
        self.body_graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.body_graph, **CAPTURE_KW):
            fr = self._body(self.state)
            self.body_stack = torch.stack(fr) if fr else None
        self._keep_b = snapshot_eager_caches()
        ops.clear_skip_proj_cache()
        fused.clear_skip_hoist_cache()

    def _warm(self):
        return trigger_warmup(*self._mods, self.x0, self.warmup, self.skip_steps)

    def _body(self, st):
        return trigger_body(st, *self._mods, self.ctx, self.coef, self.eps, self.log, self.warmup, self.total, self.probe)

    def warm(self, x0) -> None:
        self.x0.copy_(x0)
        self.warm_graph.replay()

    def run(self, index: int, eps_by_step: Optional[Dict[int, torch.Tensor]] = None) -> dict:
        """One batch index: frames (total,B,C,H,W) (static buffers: clone to keep), and on the host the trigger steps, the
        recorded values (total) and the thresholds (total - warmup)."""
        self.ctx.copy_(self.state["norms"][:, index])
        if eps_by_step is None:
            self.eps.normal_()
        else:
            for i in range(self.warmup, self.total):
                self.eps[i - self.warmup].copy_(eps_by_step[i])
        if self.body_stack is not None:
            self.body_graph.replay()
        values = torch.cat([self.state["norms"][:, index], self.log["values"][self.warmup:]]).cpu()
        thresholds = self.log["thresholds"][self.warmup:].cpu()
        flags = self.log["flags"][self.warmup:].cpu()
        frames = self.warm_stack if self.body_stack is None else torch.cat([self.warm_stack, self.body_stack])
        return {"frames": frames, "triggers": [self.warmup + int(i) for i in torch.nonzero(flags).flatten()],
                "values": [float(v) for v in values], "thresholds": [float(v) for v in thresholds],
                "values_main": self.log["values"][self.warmup:].clone()}


class GraphedRollout:
    """`sample_rollout` captured once into a hipGraph (torch.cuda.CUDAGraph) and replayed: the ~500 launches of
    a rollout (19 encoder + 10 decoder passes, 38 LSTM cells, GEMMs, the GP sample) are launch-latency-bound at
    small batch / for dcgan_64, and the per-step Python + ctypes cost disappears from the critical path.
    Inputs are copied into static buffers; the returned frames are static tensors overwritten by every replay
    (clone them to keep a sample).  The GP base samples eps (D,B) live in static buffers that every call refills from
    torch's generator on the replay stream, OUTSIDE the graph: a generator captured inside hands all graphs of a process
    the same device-side offset tensor, and two graphs replayed on different streams (ConcurrentRollouts) then read the
    same offset and draw the same sample."""

    def __init__(self, encoder, decoder, frame_predictor, gp_layer, likelihood, x, n_past, n_eval,
                 last_frame_skip=False, period=15, warmup=2):
        self._args = (encoder, decoder, frame_predictor, gp_layer, likelihood)
        self._kw = dict(n_past=n_past, n_eval=n_eval, last_frame_skip=last_frame_skip, period=period)
        buf = torch.stack([t.contiguous() for t in x])      # one buffer: the conditioning batch is a view of it
        self.static_x = [buf[i] for i in range(len(x))]
        self.eps = {i: torch.randn(gp_layer.num_dims, x[0].shape[0], device=x[0].device)
                    for i in (trigger_steps(n_past, n_eval, period) if period else [])}
        self._kw["eps_by_step"] = self.eps
        side = pooled_stream("warmup")
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):   # first-launch work (weight packs, LDS attributes, BN folds) must not be captured
                sample_rollout(*self._args, self.static_x, **self._kw)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        ops.clear_skip_proj_cache()   # nothing cached eagerly may be referenced by the graph ...
        fused.clear_skip_hoist_cache()
        with torch.cuda.graph(self.graph, **CAPTURE_KW):
            self.frames = sample_rollout(*self._args, self.static_x, **self._kw)
        self._keepalive = snapshot_eager_caches()   # eager tensors the graph reads by raw pointer live as long as it does
        ops.clear_skip_proj_cache()   # ... and nothing from the graph's pool by later eager calls
        fused.clear_skip_hoist_cache()

    def __call__(self, x: Optional[Sequence[torch.Tensor]] = None,
                 eps_by_step: Optional[Dict[int, torch.Tensor]] = None) -> List[torch.Tensor]:
        """`eps_by_step`: base samples (D,B) per trigger step (parity runs); None = fresh draws from torch's generator."""
        if x is not None:
            for dst, src in zip(self.static_x, x):
                if dst.data_ptr() != src.data_ptr():
                    dst.copy_(src)
        for i, e in self.eps.items():
            if eps_by_step is None:
                e.normal_()
            else:
                e.copy_(eps_by_step[i])
        self.graph.replay()
        return self.frames


class ConcurrentRollouts:
    """`inflight` complete rollouts of ONE batch in flight at once: the rollouts of make_gifs' `for s in range(nsample)`
    loop (generate_frames.py:143-177) are independent of each other, so each gets its own hipGraph (own static buffers, own
    private pool: no buffer is shared between two graphs; weights, packed weights and BatchNorm folds are read-only) and its
    own stream.  A rollout is a serial chain of ~500 launches, many of them small (LSTM cells, GEMV-shaped ends, the GP
    sample, the 8x8 / 4x4 layers): with several chains in flight the tail of one launch and the latency-bound phases of one
    rollout fill with another rollout's work.  Per-rollout arithmetic is unchanged (every rollout is a full
    `sample_rollout`: conditioning included, nothing shared or amortised across rollouts) and so are its results.
    `run(n)` issues n rollouts round-robin over the graphs and returns the static frame lists of the `inflight` most
    recent ones; the caller synchronises (or records events) before reading them."""

    def __init__(self, encoder, decoder, frame_predictor, gp_layer, likelihood, x, n_past, n_eval, inflight=2,
                 last_frame_skip=False, period=15, energy_tiles=None):
        if inflight < 1:
            raise ValueError("inflight must be >= 1")
        # Several chains in flight keep the board at its power cap: their graphs are captured under the ENERGY
        # tile policy (128 x 128 batched-GEMM tile, 8 x 16 pixel tiles from one workgroup per CU on: fewer LDS / L2 bytes per
        # MFMA, +4.6 % vgg_64); one chain keeps the latency tiles (`energy_tiles` overrides: profiling one chain of the in-flight
        # kernels).  The two policies' results agree to fp32 rounding, not bit for bit.
        self.energy_tiles = inflight >= 2 if energy_tiles is None else bool(energy_tiles)
        with ops.tile_policy(self.energy_tiles):
            self.rollouts = [GraphedRollout(encoder, decoder, frame_predictor, gp_layer, likelihood, x, n_past, n_eval,
                                            last_frame_skip, period) for _ in range(inflight)]
        self.streams = [pooled_stream("chain", k) for k in range(inflight)]
        self._next = 0

    def run(self, n: int, x: Optional[Sequence[torch.Tensor]] = None, chains: Optional[int] = None) -> List[List[torch.Tensor]]:
        """`chains`: use only the first `chains` graphs (1 = the rollouts back to back as one serial chain)."""
        nc = len(self.rollouts) if chains is None else max(1, min(chains, len(self.rollouts)))
        cur = torch.cuda.current_stream()
        for s in self.streams[:nc]:
            s.wait_stream(cur)            # inputs written on the caller's stream are visible to every chain
        for _ in range(n):
            k = self._next % nc
            self._next = (k + 1) % nc
            with torch.cuda.stream(self.streams[k]):
                self.rollouts[k](x)       # a stream runs its replays in order: a graph never overlaps itself
        for s in self.streams[:nc]:
            cur.wait_stream(s)
        return [r.frames for r in self.rollouts[:nc]]


class GraphedSampler:
    """`make_gifs` for one batch (generate_frames.py:107-178) as hipGraphs.  Everything that does not depend on the sample is
    replayed once per batch (three graphs: the conditioning, then side by side the other two): the conditioning (`condition`: the past frames encoded as one batch, the LSTM warmed up,
    the decoder's loop-invariant skip halves), the posterior rollout's prediction steps (`posterior_from`, :110-134 - its
    conditioning phase is the same computation as the samples') and, with `share_prefix`, the samples' prediction steps before
    the first GP trigger step with their SSIM / PSNR.  The body of the `for s in range(nsample)` loop (:163-178) - the remaining
    steps of one sample and utils.eval_seq's metrics - is one graph per chain, replayed once per sample, `inflight` samples at
    a time (one graph + one stream each, see ConcurrentRollouts); the sample graphs read the batch graph's state and
    skip-dependent tensors (read-only during their replays); GP base samples eps (D,B) per trigger step, predicted frames
    and metrics are per chain.  `set_batch()` installs a new batch, `run()` replays.

    `share_prefix` (default: rollout.SHARE_PREFIX): the samples of a batch differ only from the first GP trigger step t0 on
    (generate_frames.py:166-171: the draw at i % 15 == 0 is the loop's only source of randomness), so the prediction steps
    n_past ... t0 - 1 - the same kernels on the same inputs for every sample, like the conditioning frames - run ONCE per
    batch and every sample graph continues from the state they leave behind.  Results are bit-identical to the per-sample
    loop (tests/test_gpu_rollouts.py); with no trigger step inside the rollout all samples are the prefix."""

    def __init__(self, encoder, decoder, frame_predictor, gp_layer, likelihood, x, n_past, n_eval,
                 last_frame_skip=False, period=15, inflight=3, share_prefix=None):
        # several sample chains in flight keep the board at its power cap: their graphs are captured with the energy-lean tiles
        # (ops.tile_policy; same results to fp32 rounding); one chain at a time keeps the latency tiles
        with ops.tile_policy(max(1, inflight) >= 2):
            self._init(encoder, decoder, frame_predictor, gp_layer, likelihood, x, n_past, n_eval, last_frame_skip, period,
                       inflight, share_prefix)

    def _init(self, encoder, decoder, frame_predictor, gp_layer, likelihood, x, n_past, n_eval, last_frame_skip, period,
              inflight, share_prefix):
        self._mods = (encoder, decoder, frame_predictor, gp_layer, likelihood)
        self.n_past, self.n_eval, self.period = n_past, n_eval, period
        self.last_frame_skip = last_frame_skip
        self._kw = dict(last_frame_skip=last_frame_skip, period=period)
        dev = x[0].device
        B, D = x[0].shape[0], gp_layer.num_dims
        self.x = torch.stack([t.contiguous() for t in x])            # conditioning frames + ground truth of the metrics
        self.steps = [i for i in range(n_past, n_eval) if period and i % period == 0]
        if share_prefix is None:
            share_prefix = SHARE_PREFIX
        self.t0 = min(self.steps) if self.steps else n_eval          # first step whose outcome depends on the sample
        self.share = bool(share_prefix) and self.t0 > n_past and not last_frame_skip and n_past >= 2
        if not self.share:
            self.t0 = n_past
        self.b = None
        self.chains = []
        for k in range(max(1, inflight)):
            self.chains.append({"eps": {i: torch.zeros(D, B, device=dev) for i in self.steps},
                                "stream": pooled_stream("chain", k)})
        self.post_stream = pooled_stream("post")
        side = pooled_stream("warmup")
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):      # first-launch work (weight packs, LDS attributes, BN folds) must not be captured
            self.b = {"state": self._condition()}
            self.b["post"] = self._posterior()
            self.b["pre"] = self._prefix() if self.share else None
            self._body(self.chains[0])
            self.b = None
        torch.cuda.current_stream().wait_stream(side)
        ops.clear_skip_proj_cache()
        fused.clear_skip_hoist_cache()
        # Three graphs for the sample-independent part: the conditioning, then - independent of each other, replayed on two
        # streams - the posterior rollout's prediction steps and the samples' shared prefix.  The conditioning graph comes
        # first: the skip-dependent tensors it creates (hoisted skip halves) and those of the first decoder call after it (the
        # skip's share of the last projection) stay in the caches while the later graphs are captured, so those read buffers
        # written once per batch, before they replay, instead of recomputing them per sample.
        shared = (fused._skip_seen, fused._frozen, ops._SKIP_PROJ_CACHE)

        def capture(fn):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, **CAPTURE_KW):
                out = fn()
            return g, out, snapshot_eager_caches()

        def keys():
            return [set(c) for c in shared]

        def drop_new(keep):     # what a graph added belongs to THAT graph's pool and replays: not for the next capture
            for cache, k0 in zip(shared, keep):
                for k in [k for k in cache if k not in k0]:
                    del cache[k]

        self.b = {}
        self.cond_graph, self.b["state"], self._keep_c = capture(self._condition)
        after_cond = keys()
        # the posterior graph replays BESIDE the prefix graph (another stream): it may read what the conditioning graph wrote,
        # never what the prefix graph writes
        self.post_graph, self.b["post"], self._keep_o = capture(self._posterior)
        drop_new(after_cond)
        self.pre_graph, self.b["pre"] = None, None
        if self.share:
            self.pre_graph, self.b["pre"], self._keep_p = capture(self._prefix)
        main_stream = keys()    # written by graphs that replay on the main stream before any sample graph
        for ch in (() if self.t0 == n_eval else self.chains):
            ch["graph"], (ch["frames"], ch["ssim"], ch["psnr"]), ch["keepalive"] = capture(lambda: self._body(ch))
            drop_new(main_stream)
        ops.clear_skip_proj_cache()
        fused.clear_skip_hoist_cache()

    def _metrics(self, frames, lo, hi):
        m = [ops.eval_frames(self.x[t], frames[t]) for t in range(lo, hi)]
        return torch.stack([a for a, _ in m], 1), torch.stack([b for _, b in m], 1)

    def _condition(self) -> dict:
        """The conditioning state (rollout.condition) of the frames in self.x."""
        enc, dec, fp, gp, lik = self._mods
        return condition(enc, fp, [self.x[i] for i in range(self.x.shape[0])], self.n_past, self.last_frame_skip, decoder=dec)

    def _posterior(self) -> torch.Tensor:
        return torch.stack(posterior_from(self.b["state"], *self._mods, self.n_past, self.n_eval, self.last_frame_skip))

    def _prefix(self) -> dict:
        """The samples' steps n_past ... t0 - 1 from the conditioning state: frames, their metrics, the LSTM state after them."""
        st = self.b["state"]
        frames = sample_from(st, *self._mods, n_past=self.n_past, n_eval=self.t0, **self._kw)
        ssim, psnr = self._metrics(frames, self.n_past, self.t0)
        return {"hidden": list(self._mods[2].hidden), "skip": st["skip"], "frames": frames, "stack": torch.stack(frames),
                "ssim": ssim, "psnr": psnr}

    def _body(self, ch):
        """One sample from step t0 on: (frames t0 ... n_eval - 1 stacked - with the conditioning frames in front when there is
        no shared prefix -, SSIM, PSNR of the predicted ones among them)."""
        if self.t0 == self.n_eval:       # no trigger step inside the rollout: every sample IS the prefix
            return None, None, None
        state = self.b["pre"] if self.share else self.b["state"]
        frames = sample_from(state, *self._mods, eps_by_step=ch["eps"], n_past=self.t0, n_eval=self.n_eval, **self._kw)
        ssim, psnr = self._metrics(frames, self.t0, self.n_eval)
        return torch.stack(frames[self.t0 if self.share else 0:]), ssim, psnr

    def set_batch(self, x) -> None:
        """New frames (conditioning + ground truth), copied into the static buffer on the current stream."""
        for i, t in enumerate(x):
            self.x[i].copy_(t)

    def run(self, nsample: int, samples: torch.Tensor, ssim: torch.Tensor, psnr: torch.Tensor,
            eps_by_sample: Optional[Sequence[Dict[int, torch.Tensor]]] = None) -> torch.Tensor:
        """Draws `nsample` samples: samples[s] <- the n_eval frames (n_eval,B,C,H,W), ssim[:, s] / psnr[:, s] <- (B,T); returns
        the posterior rollout's frames (n_eval,B,C,H,W) - a static buffer the next run() overwrites.
        eps_by_sample[s][i]: base sample of sample s at trigger step i (parity runs); None = torch's generator."""
        cur = torch.cuda.current_stream()
        P = self.t0 - self.n_past
        self.cond_graph.replay()             # on the current stream: everything below waits for it
        self.post_stream.wait_stream(cur)
        with torch.cuda.stream(self.post_stream):
            self.post_graph.replay()         # beside the prefix and the samples (it only reads the conditioning state)
        if self.share:
            self.pre_graph.replay()
            pre = self.b["pre"]
            samples[:, :self.t0].copy_(pre["stack"].unsqueeze(0).expand(nsample, *pre["stack"].shape))
            ssim[:, :, :P].copy_(pre["ssim"].unsqueeze(1).expand(-1, nsample, -1))
            psnr[:, :, :P].copy_(pre["psnr"].unsqueeze(1).expand(-1, nsample, -1))
            if self.t0 == self.n_eval:
                cur.wait_stream(self.post_stream)
                return self.b["post"]
        for ch in self.chains:
            ch["stream"].wait_stream(cur)
        lo = self.t0 if self.share else 0
        for s in range(nsample):
            ch = self.chains[s % len(self.chains)]
            with torch.cuda.stream(ch["stream"]):
                for i in self.steps:
                    if eps_by_sample is None:
                        ch["eps"][i].normal_()
                    else:
                        ch["eps"][i].copy_(eps_by_sample[s][i])
                ch["graph"].replay()
                samples[s, lo:].copy_(ch["frames"])
                ssim[:, s, P:].copy_(ch["ssim"])
                psnr[:, s, P:].copy_(ch["psnr"])
        for ch in self.chains:
            cur.wait_stream(ch["stream"])
        cur.wait_stream(self.post_stream)
        return self.b["post"]
