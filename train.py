#!/usr/bin/env python3
"""train.py — counterpart of the reference's train.py on the MI355X kernel library.

Same flags (train.py:17-46) and the same three step closures (`train_model` :200-248,
`train_frame_predictor` :175-198, `train_GP_Frame_predictor` :146-172), loss weights (:239), optimisers
(:95-106, lr hard-coded 0.002 like the reference; `--lr/--beta1/--optimizer/--z_dim/--name` stay accepted
and unused), scheduler-before-epoch order (:347), log line (:368) and checkpoint dict (:380-388).

Consciously fixed (each was unrunnable in the reference — SURVEY.md §5 quirks):
  * the model family follows `--model` / `--image_width` instead of a hard-coded dcgan_64 (:75);
  * `h.view(90, 50, 1)` (:164) uses `(g_dim, batch_size, 1)`;
  * `--dataset smmnist` works (`--num_digits`), batches may be `x` or `(x, y)`;
  * no `torch.cuda.empty_cache()` per timestep (:166,191,235);
  * `--ft` / flags are real booleans (`--no_ft`).
Reference behaviour that is KEPT although it looks like a bug, with a switch (each has a test showing both modes):
  * `Trainer.reference_gp_grad_leak = True`: train_model (:200-245) zeroes the encoder / decoder / LSTM gradients but
    not the GP optimiser's, so the full-scale ELBO gradients that the previous iteration's train_GP_Frame_predictor
    left in `.grad` (:170-171) are still there when train_model's `optimizer.step()` (:245) runs, on top of the
    1e-4-weighted ones of :239.  False zeroes them first.
Added: data parallelism — launch with `python -m torch.distributed.run --nproc-per-node N train.py ...`;
`--batch_size` is the GLOBAL batch, split evenly over the ranks (the ELBO's `num_data` is the global batch as well, so
the KL weight does not change with the number of GPUs); every parameter's `.grad` is a view of one flat arena
(dvg_amd/optim.py) whose ranges are averaged in place over RCCL, the decoder-side range while the encoder phase of the
backward pass is still running (dvg_amd/parallel.py); BatchNorm statistics are per replica.
Datasets are synthetic here (no network / files offline): `smmnist` = seeded Moving-MNIST trajectories with in-repo
sprites; any other `--dataset` needs `--synthetic_data` (random textured clips of the right shape) and fails without
it, because `--data_root` cannot be honoured.
"""
import argparse
import importlib
import os
import random
import sys
import time

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import utils  # noqa: E402
from dvg_amd import fused, parallel  # noqa: E402
from dvg_amd.data import SyntheticMovingMNIST, make_batch_generator, synthetic_video  # noqa: E402,F401
from dvg_amd.models.gp_models import GaussianLikelihood, GPRegressionLayer1, VariationalELBO  # noqa: E402
from dvg_amd.optim import FlatArena, FusedAdam, zero_grads  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--lr', default=0.002, type=float, help='learning rate (unused, as in the reference)')
    p.add_argument('--beta1', default=0.9, type=float)
    p.add_argument('--batch_size', default=50, type=int, help='GLOBAL batch size')
    p.add_argument('--log_dir', default='logs')
    p.add_argument('--model_dir', default='')
    p.add_argument('--name', default='')
    p.add_argument('--output_path', default='.')
    p.add_argument('--data_root', default='path/to/data/')
    p.add_argument('--optimizer', default='adam')
    p.add_argument('--niter', type=int, default=601)
    p.add_argument('--seed', default=1, type=int)
    p.add_argument('--epoch_size', type=int, default=300)
    p.add_argument('--image_width', type=int, default=64)
    p.add_argument('--channels', default=1, type=int)
    p.add_argument('--dataset', default='smmnist')
    p.add_argument('--num_digits', type=int, default=2)
    p.add_argument('--n_past', type=int, default=5)
    p.add_argument('--no_ft', action='store_true', help='disable the temporal fine-tuning closures')
    p.add_argument('--n_future', type=int, default=10)
    p.add_argument('--n_eval', type=int, default=15)
    p.add_argument('--rnn_size', type=int, default=256)
    p.add_argument('--predictor_rnn_layers', type=int, default=2)
    p.add_argument('--z_dim', type=int, default=10)
    p.add_argument('--g_dim', type=int, default=90)
    p.add_argument('--model', default='dcgan', help='dcgan | vgg')
    p.add_argument('--data_threads', type=int, default=5)
    p.add_argument('--last_frame_skip', action='store_true')
    p.add_argument('--save_every', type=int, default=4)
    p.add_argument('--no_save', action='store_true')
    p.add_argument('--hip_graph', action='store_true',
                   help='(the default) replay each training iteration as one hipGraph (GraphedIteration; same losses and '
                        'parameters as the eager loop, tested; 1.5-2.7x the train frames/s); with more than one rank: as a '
                        'chain of hipGraphs cut at the gradient all-reduces, which stay eager (SegmentedIteration)')
    p.add_argument('--no_hip_graph', action='store_true', help='eager launches for every iteration')
    p.add_argument('--print_param_checksum', action='store_true',
                   help='print a checksum of all parameters per rank at the end (multi-rank tests: every rank must agree)')
    p.add_argument('--sync_bn', action='store_true',
                   help='several ranks: BatchNorm statistics (and their backward sums) over the GLOBAL batch - one all-reduce of '
                        '2 x C floats per BatchNorm call and direction - i.e. the reference\'s single-process semantics: N ranks x '
                        'B/N clips then train exactly like one process on B clips (default: per-replica statistics, as '
                        'DistributedDataParallel).  The collectives cannot be captured in a hipGraph: iterations run eager.')
    p.add_argument('--synthetic_data', action='store_true',
                   help='datasets other than smmnist: train on synthetic clips of the right shape (--data_root is not read)')
    return p


class Trainer:
    """The module-level state of the reference script, as an object (so tests can drive single steps)."""

    def __init__(self, opt, device):
        self.opt, self.dev = opt, device
        self.rank, self.world = opt.rank, opt.world
        model = importlib.import_module(f"models.{opt.model}_{opt.image_width}")
        import models.lstm as lstm_models
        self.encoder = model.encoder(opt.g_dim, opt.channels)
        self.decoder = model.decoder(opt.g_dim, opt.channels)
        self.encoder.apply(utils.init_weights)
        self.decoder.apply(utils.init_weights)
        self.frame_predictor = lstm_models.lstm(opt.g_dim, opt.g_dim, opt.rnn_size, opt.predictor_rnn_layers,
                                                opt.local_batch)
        self.frame_predictor.apply(utils.init_weights)
        self.gp_layer = GPRegressionLayer1(num_dims=opt.g_dim)
        self.likelihood = GaussianLikelihood(batch_size=opt.g_dim)
        self.modules = [self.encoder, self.decoder, self.frame_predictor, self.gp_layer, self.likelihood]
        for m in self.modules:
            m.to(device)
        # True = the reference's behaviour: train_model does not zero the GP optimiser's gradients (see module docstring)
        self.reference_gp_grad_leak = True
        # True = backward in two phases (decoder / LSTM / GP side, then the encoder) so that the all-reduce of the first
        # phase's gradients overlaps the second phase; needs share_encoder_passes
        self.staged_backward = True
        # True = back-propagate into the encoder in the two fine-tuning closures like the reference does (and then
        # discards); kept only so that tests can show both ways give the same updates
        self.finetune_encoder_grad = False
        # False = encode every middle frame twice per closure like the reference does (tests compare both ways)
        self.share_encoder_passes = True
        # True = the three decoder calls of a time step compute the skip half of each concat conv once (forward + backward)
        self.share_skip_halves = True
        # True = the GP fine-tuning closure reuses the encodings of the LSTM fine-tuning closure that precedes it
        self.share_closure_encodings = True
        # True = the encoder calls of a closure run as ONE pass over all frames with per-frame ("grouped") BatchNorm
        # statistics (needs share_encoder_passes); DVG_TIME_BATCH=0: one pass per frame
        self.time_batched = os.environ.get("DVG_TIME_BATCH", "2") != "0"
        # True (with time_batched) = train_model's 3 S decoder calls as one decoder pass with shared skip blocks
        # (_train_model_batched); DVG_TIME_BATCH=1: encoder only
        self.time_batched_decoder = os.environ.get("DVG_TIME_BATCH", "2") not in ("0", "1")
        # True (with time_batched) = the LSTM of a teacher-forced closure over the whole sequence at once: one GEMM per
        # non-recurrent product over all S x B rows, one launch per step and layer for the recurrence (models.lstm.
        # forward_sequence); lstm_sequence = False: one module call per step (an environment switch until r06)
        self.lstm_sequence = True
        # True = train_model's latent path (LSTM, GP, latent losses) on a second stream, concurrent with the decoder calls
        self.latent_stream = True
        # True = the closures' losses and their gradients by dvg_frame_losses / dvg_mse_sum_grad, backward seeded with them
        # (no scalar autograd graph of torch ops); False: the torch composition (tests compare the two)
        self.fused_losses = True
        self._loss_w = {}
        self._side_stream = None
        # optim.Adam(lr=0.002) x4 (train.py:95-104) as one fused HIP launch per parameter group.  All groups live in ONE
        # flat arena in the order [GP | likelihood | LSTM | decoder | encoder]: parameters, gradients (p.grad are views)
        # and both moments; the data-parallel all-reduce works on ranges of arena.g in place.
        Adam = FusedAdam   # HIP only, like the models themselves: no CPU fallback on the product path
        allp = [p for m in self.modules for p in m.parameters()]
        self.arena = FlatArena(FlatArena.size_for(allp), device)
        self.optimizer = Adam([{'params': self.gp_layer.parameters()},
                               {'params': self.likelihood.parameters()}], lr=0.002, arena=self.arena)
        self.frame_predictor_optimizer = Adam(self.frame_predictor.parameters(), lr=0.002, arena=self.arena)
        self.decoder_optimizer = Adam(self.decoder.parameters(), lr=0.002, arena=self.arena)
        self.encoder_optimizer = Adam(self.encoder.parameters(), lr=0.002, arena=self.arena)
        self.scheduler = torch.optim.lr_scheduler.MultiStepLR(self.optimizer, milestones=[3, 5], gamma=0.1)
        # num_data = the GLOBAL batch (train.py:112 passes opt.batch_size): each rank's loss is ll_r/B_local - KL/num_data and
        # the ranks are averaged, which gives the reference's ll/B - KL/B at the same global batch for any number of ranks
        self.mll = VariationalELBO(self.likelihood, self.gp_layer, num_data=opt.batch_size, combine_terms=True)
        self.mse_criterion = nn.MSELoss()
        self.mse_latent_criterion = nn.MSELoss()
        # every replica starts from rank 0's values: ONE broadcast of the parameter arena (all parameters are views of it by
        # now) + one packed broadcast per buffer dtype, instead of ~200 per-tensor broadcasts
        self.broadcast_collectives = parallel.broadcast_parameters(self.modules, arena_p=self.arena.p)
        # --sync_bn: BatchNorm over the global batch (fused.set_sync_bn).  Its small collectives get their OWN process group
        # (own RCCL communicator and stream): they must not queue behind the gradient ranges that are in flight while the
        # encoder phase of the backward pass - which issues them - is still running
        self.sync_bn = False
        if getattr(opt, 'sync_bn', False) and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            fused.set_sync_bn(torch.distributed, torch.distributed.new_group())
            self.sync_bn = True
        self.reducer = parallel.ArenaReducer(self.arena.g)
        self.rng_gp = (self.optimizer.flat_range(0)[0], self.optimizer.flat_range(1)[1])
        self.rng_fp = self.frame_predictor_optimizer.flat_range(0)
        self.rng_dec = self.decoder_optimizer.flat_range(0)
        self.rng_enc = self.encoder_optimizer.flat_range(0)
        assert self.rng_gp[1] == self.rng_fp[0] and self.rng_fp[1] == self.rng_dec[0] and self.rng_dec[1] == self.rng_enc[0]

    # ---- data-parallel switches / statistics (bench.py's training leg) --------------------------
    def _ar(self, *actions):
        """Gradient all-reduce actions at this point of the iteration: ("reduce", (lo, hi)) | ("start", key, (lo, hi)) |
        ("finish", key) on ranges of the flat gradient arena.  Normally run at once; while a SegmentedIteration captures,
        the graph is cut here instead and the actions are replayed eagerly between the graph segments."""
        seg = getattr(self, "_segmenter", None)
        if seg is not None:
            if self.reducer.active():
                seg.cut(actions)
            return
        self._run_ar(actions)

    def _run_ar(self, actions):
        pend = self.__dict__.setdefault("_ar_pending", {})
        for a in actions:
            if a[0] == "reduce":
                self.reducer.reduce(*a[1])
            elif a[0] == "start":
                pend[a[1]] = self.reducer.start(*a[2])
            else:
                self.reducer.finish(pend.pop(a[1], None))

    def set_allreduce(self, on: bool):
        self.reducer.enabled = bool(on)

    def reset_allreduce_stats(self):
        self.reducer.calls = self.reducer.floats = 0
        self._iters = 0

    def allreduce_stats(self):
        """Collectives and bytes per iteration, and their standalone duration (the same ranges all-reduced back to back,
        timed with events on the launch stream; how much of it an iteration actually waits for is a separate measurement)."""
        if not self.reducer.active() or not getattr(self, '_iters', 0):
            return None
        import torch.distributed as dist
        ranges = [(self.rng_gp[0], self.rng_dec[1]), self.rng_enc, (self.rng_gp[0], self.rng_fp[1])] if self.opt.ft else \
            [(self.rng_gp[0], self.rng_dec[1]), self.rng_enc]
        scratch = torch.zeros_like(self.arena.g)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(3):
            for lo, hi in ranges:
                dist.all_reduce(scratch[lo:hi])
        torch.cuda.synchronize()
        reps = 10
        e0.record()
        for _ in range(reps):
            for lo, hi in ranges:
                dist.all_reduce(scratch[lo:hi])
        e1.record()
        torch.cuda.synchronize()
        return {"allreduce_ms_per_iter": round(e0.elapsed_time(e1) / reps, 3),
                "allreduces_per_iter": round(self.reducer.calls / self._iters, 2),
                "allreduce_MB_per_iter": round(4e-6 * self.reducer.floats / self._iters, 2)}

    # ---- mode switches (train.py:342-346,372-374) -------------------------------------------
    def train_mode(self):
        for m in self.modules:
            m.train()

    def _latent_stream(self):
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream()
            from dvg_amd import autograd as _ag
            _ag.JOIN_STREAMS.append(self._side_stream)   # the end-of-backward weight-gradient flush waits for it
        return self._side_stream

    def _lstm_seq_applies(self):
        from dvg_amd.models.lstm import sequence_applies
        return (self.lstm_sequence and self.time_batched and self.frame_predictor.training and torch.is_grad_enabled()
                and sequence_applies(self.frame_predictor))

    def _gp_in(self, h):
        return h.transpose(0, 1).view(self.opt.g_dim, h.shape[0], 1)

    def _skip_rule(self, i, enc_out, skip):
        """train.py:158-161,184-187,217-220: skip is refreshed only while i < n_past (or last_frame_skip)."""
        if self.opt.last_frame_skip or i < self.opt.n_past:
            return enc_out[0], enc_out[1]
        return enc_out[0], skip

    # ---- the three closures -------------------------------------------------------------------
    def _encode_sequence(self, x, grad: bool):
        """encoder(x[t]) for every frame of the sequence, ONCE per frame.  The reference encodes every middle frame twice
        per closure (train.py:158-162,184-188,217-221: as step i's target and again as step i+1's input) with the same
        weights and the same batch, i.e. with identical outputs; one pass whose output is used twice gives the same
        losses and - autograd sums the two uses - the same gradients, and `fused.bn_passes(2)` reproduces the BatchNorm
        side effect of the second pass (two momentum updates, num_batches_tracked += 2) in the reference's order.
        (6S encoder passes per iteration become 3(S+1).)"""
        T = self.opt.n_past + self.opt.n_future
        if self.time_batched and self.encoder.training and T >= 3:
            return self._encode_sequence_batched(x, grad, T)
        outs = []
        with torch.set_grad_enabled(grad):
            for t in range(T):
                with fused.bn_passes(1 if (t == 0 or t == T - 1) else 2):
                    outs.append(self.encoder(x[t]))
        return outs

    def _encode_sequence_batched(self, x, grad: bool, T: int):
        """The T encoder calls of a closure as ONE pass over T x B frames.  train.py:213-221 is teacher-forced - every call
        encodes a ground-truth frame - so the calls are independent of each other; what ties a reference call together is
        its BatchNorm batch, and that is kept: statistics, normalisation and their backward are per GROUP of B consecutive
        images (one group per frame), and the running statistics advance frame by frame in the reference's order with the
        pass counts of the per-frame path (first / last frame once, middle frames twice).  Same losses, gradients and
        buffers up to fp32 summation order (tests/test_gpu_train.py); every conv launch is T x larger."""
        from dvg_amd.autograd import split_batch
        from dvg_amd.rollout import _adjacent_view
        frames = _adjacent_view(list(x[:T]))
        if frames is None:
            frames = torch.cat(list(x[:T]), 0)
        with torch.set_grad_enabled(grad), fused.bn_groups(T, (1, 2, 1)):
            h_all, skips_all = self.encoder(frames)
        hs = split_batch(h_all, T)
        sks = [split_batch(s, T) for s in skips_all]
        return [(hs[t], [sk[t] for sk in sks]) for t in range(T)]

    def _share_scope(self):
        import contextlib
        return fused.share_skip_halves() if self.share_skip_halves else contextlib.nullcontext()

    def _enc(self, enc_all, x, t, grad=True):
        if enc_all is not None:
            return enc_all[t]
        with torch.set_grad_enabled(grad):
            return self.encoder(x[t])

    # Each closure has a `_dev` form that returns device tensors and never synchronises with the host, so that a whole
    # iteration can be captured into one hipGraph (GraphedIteration below); the public forms add the float()s.
    def train_GP_Frame_predictor(self, x):
        return float(self._train_gp_dev(x)) / (self.opt.n_past + self.opt.n_future)

    def _loss_weights(self, dev, n_call, D):
        """[0.001 / n, 1000 / n, 0.001 / n, 0.01, -0.0001 x D]: train.py:239's weights on (sum sq mse, ae_mse, mse_gp, mse_latent,
        the D = S x g_dim entries of elbo) - max_ll = -elbo -; its tail [4:] is d loss / d elbo."""
        key = (dev, n_call, D)
        w = self._loss_w.get(key)
        if w is None:
            w = torch.tensor([0.001 / n_call, 1000.0 / n_call, 0.001 / n_call, 0.01] + [-0.0001] * D, dtype=torch.float32, device=dev)
            if not torch.cuda.is_current_stream_capturing():    # a tensor made during a capture lives in the graph's pool
                self._loss_w[key] = w                           # (ADVICE r05): never handed to later eager iterations
        return w

    def train_frame_predictor(self, x):
        return float(self._train_fp_dev(x)) / (self.opt.n_past + self.opt.n_future)

    def train_model(self, x):
        mse_latent, loss = self._train_model_dev(x)
        v = float(mse_latent) / (self.opt.n_past + self.opt.n_future)
        self.last_loss = float(loss)
        return v, v

    def _train_gp_dev(self, x, defer_step=False):
        opt = self.opt
        zero_grads([self.optimizer])
        self.frame_predictor.hidden = None      # train.py:150 re-creates it here, but this closure never steps the LSTM
        max_ll = 0
        skip = None
        g = self.finetune_encoder_grad
        cache, self._ft_cache = getattr(self, '_ft_cache', None), None
        if (cache is not None and cache[0] is x and not g and self.encoder.training and
                cache[3] == tuple(p._version for p in self.encoder.parameters())):
            enc_all = cache[1]                       # same frames, same encoder weights: identical encodings ...
            fused.replay_bn_trace(cache[2])          # ... and the BatchNorm side effects of re-encoding them, replayed
        else:
            enc_all = self._encode_sequence(x, g) if self.share_encoder_passes else None
        if self.time_batched and enc_all is not None and not g and self.gp_layer.training:
            # no recurrence in this closure: all S GP posteriors + ELBO terms as one launch (gp_autograd.gp_elbo_steps)
            from dvg_amd.gp_autograd import gp_elbo_steps
            hcat = torch.stack([e[0].detach() for e in enc_all])          # (T, B, D)
            elbo, _ = gp_elbo_steps(self.gp_layer, self.mll, hcat[:-1], hcat[1:])
            if self.fused_losses:     # loss = (-elbo).sum(): d loss / d elbo = -1, no scalar graph
                gneg = self._loss_w.get(("neg1", elbo.device, elbo.numel()))
                if gneg is None:
                    gneg = torch.full((elbo.numel(),), -1.0, device=elbo.device)
                    if not torch.cuda.is_current_stream_capturing():   # as autograd._zero_state: a tensor first made during a
                        self._loss_w[("neg1", elbo.device, elbo.numel())] = gneg   # capture is written only when that graph replays
                loss = torch.dot(elbo.detach(), gneg)
                elbo.backward(gneg)
                if not defer_step:
                    self._ar(("reduce", self.rng_gp))
                    self.optimizer.step()
                return loss
            max_ll = -elbo
        else:
            for i in range(1, opt.n_past + opt.n_future):
                # Only the GP optimiser steps after this closure (train.py:170-171): the reference back-propagates into
                # the encoder and then discards those gradients (encoder.zero_grad() opens the next train_model).  The
                # encoder therefore runs without autograd here - same outputs, same BatchNorm running-stat updates, same
                # parameter updates, none of the wasted encoder backward.
                h, skip = self._skip_rule(i, self._enc(enc_all, x, i - 1, g), skip)
                h_target = self._enc(enc_all, x, i, g)[0].detach()
                h_pred = self.gp_layer(self._gp_in(h))
                max_ll = max_ll - self.mll(h_pred, h_target.transpose(0, 1))
        loss = max_ll.sum()
        loss.backward()
        if not defer_step:
            self._ar(("reduce", self.rng_gp))
            self.optimizer.step()
        return loss.detach()

    def _train_fp_dev(self, x, defer_step=False):
        opt = self.opt
        zero_grads([self.frame_predictor_optimizer])   # frame_predictor.zero_grad() (train.py:176): one fill of the flat range
        self.frame_predictor.hidden = None               # (re-created below by the step-by-step path; the sequence form starts
        #                                                   from its own zero state and leaves None behind)
        mse_latent = 0
        skip = None
        g = self.finetune_encoder_grad   # only frame_predictor_optimizer steps (train.py:195-196)
        self._ft_cache = None
        if self.share_encoder_passes and self.share_closure_encodings and not g:
            # the GP closure that follows encodes the same frames with the same encoder weights: keep the encodings and
            # the BatchNorm statistics of these passes for it (fused.bn_trace / replay_bn_trace)
            with fused.bn_trace() as trace:
                enc_all = self._encode_sequence(x, False)
            self._ft_cache = (x, enc_all, trace.entries, tuple(p._version for p in self.encoder.parameters()))
        else:
            enc_all = self._encode_sequence(x, g) if self.share_encoder_passes else None
        if self._lstm_seq_applies() and enc_all is not None and not g:
            # teacher-forced: all S inputs exist before the recurrence starts - the sequence in one pass
            from dvg_amd.models.lstm import forward_sequence
            hcat = torch.stack([e[0].detach() for e in enc_all])              # (T, B, D)
            pred = forward_sequence(self.frame_predictor, hcat[:-1])
            if self.fused_losses:     # the sum of squares and its gradient in one launch; backward starts from d mse / d pred
                from dvg_amd import ops
                sq, d_pred = ops.mse_sum_grad(pred.detach(), hcat[1:], 1.0 / float(hcat[0].numel()))
                mse_latent = sq / float(hcat[0].numel())
                pred.backward(d_pred)
                if not defer_step:
                    self._ar(("reduce", self.rng_fp))
                    self.frame_predictor_optimizer.step()
                return mse_latent
            d = pred - hcat[1:]
            mse_latent = (d * d).sum() / float(hcat[0].numel())                # sum over the steps of nn.MSELoss (mean)
        else:
            self.frame_predictor.hidden = self.frame_predictor.init_hidden()       # train.py:178
            for i in range(1, opt.n_past + opt.n_future):
                h, skip = self._skip_rule(i, self._enc(enc_all, x, i - 1, g), skip)
                h_target = self._enc(enc_all, x, i, g)[0]
                h_pred = self.frame_predictor(h)
                mse_latent = mse_latent + self.mse_latent_criterion(h_pred, h_target)
        mse_latent.backward()
        if not defer_step:
            self._ar(("reduce", self.rng_fp))
            self.frame_predictor_optimizer.step()
        return mse_latent.detach()

    def _train_model_batched(self, x):
        """train_model (train.py:200-248) with the encoder AND decoder calls time-batched.  The closure is teacher-forced:
        step i reads the encodings of the ground-truth frames x[i-1], x[i]; so (1) all T frames are encoded in one pass
        (grouped BatchNorm, _encode_sequence_batched); (2) the latent chain - LSTM step, GP posterior + ELBO term, latent
        MSE, the only part with a recurrence - runs step by step on slices of those encodings; (3) the 3 S decoder calls
        (x_pred, x_target_pred, x_pred_gp for every step, :227-232) run as ONE decoder pass over 3 S B latents in the
        reference's call order, BatchNorm per call (group), with the skip tensors as SHARED BLOCKS: the three calls of a
        step read the same skip, and from step n_past on the skip is frozen (:217-220), so the skip half of every concat
        conv is computed (and back-propagated) once per DISTINCT skip - n_past - 1 times, not 3 S times.  Losses, gradients
        and buffers equal the step-by-step path up to fp32 summation order (tests/test_gpu_train.py)."""
        from dvg_amd import ops
        from dvg_amd.autograd import split_batch
        from dvg_amd.rollout import _adjacent_view
        opt = self.opt
        T = opt.n_past + opt.n_future
        S = T - 1
        B = x[0].shape[0]
        # encoder / decoder / frame_predictor .zero_grad() (train.py:201-203) - adjacent ranges of the gradient arena: one fill
        zero_grads([self.encoder_optimizer, self.decoder_optimizer, self.frame_predictor_optimizer] +
                   ([] if self.reference_gp_grad_leak else [self.optimizer]))
        self.frame_predictor.hidden = None    # (the step-by-step branch below re-creates it; the sequence form never reads it)
        frames = _adjacent_view(list(x[:T]))
        if frames is None:
            frames = torch.cat(list(x[:T]), 0)
        with fused.bn_groups(T, (1, 2, 1)):
            h_all, skips_all = self.encoder(frames)
        staged = self.staged_backward
        if staged:     # cut the graph at the encoder outputs (see _train_model_dev)
            h_leaf = h_all.detach().requires_grad_(True)
            sk_leaf = [s.detach().requires_grad_(True) for s in skips_all]
        else:
            h_leaf, sk_leaf = h_all, list(skips_all)
        hs = split_batch(h_leaf, T)
        # the GP posterior + ELBO term of all S steps (no recurrence there: step i reads h(x[i-1]), h(x[i])): one launch
        from dvg_amd.gp_autograd import gp_elbo_steps
        D = h_leaf.shape[1]
        elbo, gp_means = gp_elbo_steps(self.gp_layer, self.mll, h_leaf[:S * B].view(S, B, D), h_leaf[B:].view(S, B, D))
        max_ll = -elbo
        # latent chain (the only recurrence): the LSTM steps
        fused_losses = self._lstm_seq_applies() and self.fused_losses
        if self._lstm_seq_applies():
            from dvg_amd.models.lstm import forward_sequence
            tgt_h = h_leaf[B:].view(S, B, D)
            pred = forward_sequence(self.frame_predictor, h_leaf[:S * B].view(S, B, D))      # (S, B, D)
            if fused_losses:     # sum of squares + its gradient in one launch (the scalar graph of :223,239 is not built)
                lat_sq, d_pred = ops.mse_sum_grad(pred.detach(), tgt_h.detach(), 0.01 / float(B * D))
                mse_latent = lat_sq / float(B * D)
            else:
                dlat = pred - tgt_h
                mse_latent = (dlat * dlat).sum() / float(B * D)                                  # sum over the steps of nn.MSELoss
            vec_all = torch.stack([pred, tgt_h, gp_means], 1).reshape(3 * S * B, D)              # reference call order
        else:
            self.frame_predictor.hidden = self.frame_predictor.init_hidden()       # train.py:206
            mse_latent = 0
            vecs = []
            for i in range(1, T):
                h, h_target = hs[i - 1], hs[i]
                h_pred = self.frame_predictor(h)
                mse_latent = mse_latent + self.mse_latent_criterion(h_pred, h_target)
                vecs += [h_pred, h_target, gp_means[i - 1]]
            vec_all = torch.cat(vecs, 0)                              # (3 S B, g_dim), reference call order
        # skip of step i: the encoder's skips of frame i-1 while i < n_past (or always with last_frame_skip), frozen after
        nblk = S if opt.last_frame_skip else max(1, opt.n_past - 1)
        gmap = tuple(min(g // 3, nblk - 1) for g in range(3 * S))
        mdev = ops.shared_map(gmap, vec_all.device)
        shared = [ops.SharedBlocks(s[:nblk * B], B, mdev, gmap) for s in sk_leaf]
        with fused.bn_groups(3 * S, (1, 1, 1)):
            x_all = self.decoder([vec_all, shared])               # (3 S B, nc, H, W)
        if fused_losses:
            # loss = 1000 ae_mse + 0.001 mse + 0.01 mse_latent + 0.001 mse_gp + 0.0001 max_ll.sum() (train.py:239) WITHOUT its
            # scalar autograd graph: the three frame terms' sums of squares and d loss / d x_all in one pass over the 3 S decoder
            # outputs (ops.frame_losses; call order x_pred, x_target_pred, x_pred_gp = mse, ae_mse, mse_gp), the latent term's
            # above, d loss / d elbo = -0.0001; backward starts from those seeds.  ~40 elementwise / reduction launches of torch
            # per iteration (two of them passes over all 3 S B frames) become 3.
            n_call = float(frames[0].numel() * B)
            sq, d_x = ops.frame_losses(x_all.detach().view(S, 3, -1), frames[B:].view(S, -1),
                                       (0.001 / n_call, 1000.0 / n_call, 0.001 / n_call))
            wv = self._loss_weights(x_all.device, n_call, elbo.numel())       # elbo: one entry per (step, latent dim)
            loss = torch.dot(torch.cat([sq, mse_latent.view(1), elbo.detach()]), wv)
            torch.autograd.backward([x_all, pred, tgt_h, elbo], [d_x.view_as(x_all), d_pred, -d_pred, wv[4:]])
        else:
            tgt = frames[B:].view(S, 1, B, *frames.shape[1:])
            per = x_all.view(S, 3, B, *frames.shape[1:]) - tgt
            # nn.MSELoss per call, summed over the steps = sum of squares / elements per call
            sq = (per * per).sum((0, 2, 3, 4, 5)) / float(per[0, 0].numel())
            mse, ae_mse, mse_gp = sq[0], sq[1], sq[2]
            loss = 1000 * ae_mse + 0.001 * mse + 0.01 * mse_latent + 0.001 * mse_gp + 0.0001 * max_ll.sum()
            loss.backward()
        if staged:
            self._ar(("start", "a", (self.rng_gp[0], self.rng_dec[1])))
            outs, seeds = [h_all], [h_leaf.grad]
            for s, l in zip(skips_all, sk_leaf):
                if l.grad is not None:
                    outs.append(s)
                    seeds.append(l.grad)
            torch.autograd.backward(outs, seeds)
            self._ar(("start", "b", self.rng_enc), ("finish", "a"), ("finish", "b"))
        else:
            self._ar(("reduce", (self.rng_gp[0], self.rng_enc[1])))
        self.frame_predictor_optimizer.step()
        self.encoder_optimizer.step()
        self.decoder_optimizer.step()
        self.optimizer.step()
        return mse_latent.detach(), loss.detach()

    def _train_model_dev(self, x):
        opt = self.opt
        if (self.time_batched and self.time_batched_decoder and self.share_encoder_passes and self.encoder.training
                and self.decoder.training and opt.n_past >= 2 and opt.n_past + opt.n_future >= 3):
            return self._train_model_batched(x)
        self.encoder_optimizer.zero_grad()            # encoder / decoder / frame_predictor .zero_grad() (train.py:201-203)
        self.decoder_optimizer.zero_grad()
        self.frame_predictor_optimizer.zero_grad()
        if not self.reference_gp_grad_leak:
            self.optimizer.zero_grad()                # the reference does NOT do this (see the module docstring)
        self.frame_predictor.hidden = self.frame_predictor.init_hidden()
        mse = mse_latent = mse_gp = ae_mse = 0
        max_ll = 0
        skip = None
        enc_all = self._encode_sequence(x, True) if self.share_encoder_passes else None
        staged = enc_all is not None and self.staged_backward
        if staged:
            # cut the graph at the encoder outputs: the decoder / LSTM / GP phase back-propagates into detached leaves,
            # whose gradients then seed the encoder phase (same sums, same gradients)
            enc_out = enc_all
            enc_all = [(h.detach().requires_grad_(True), [s.detach().requires_grad_(True) for s in sk])
                       for h, sk in enc_out]
        # The latent path (LSTM step, GP posterior + ELBO term, latent MSE: a few dozen small latency-bound kernels per
        # step) runs on a second stream, ahead of the decoder calls it feeds: the decoders' MFMA-bound kernels hide it.
        # autograd replays the stream assignment in the backward pass (a node runs on its forward stream, with event
        # dependencies); the end-of-backward weight-gradient flush and this function join the streams explicitly.
        cur = torch.cuda.current_stream()
        # (eager iterations only: replayed as a hipGraph the two-stream DAG measured 3-5 ms SLOWER per iteration - the
        # cross-stream dependency edges cost more than the overlap of kernels that no longer wait for the CPU gives; also at
        # the small per-GPU batches of the data-parallel shapes: 32.9 -> 35.3 ms (C4 dcgan_64), 50.4 -> 54.4 (C5 dcgan_128))
        side = self._latent_stream() if self.latent_stream and not torch.cuda.is_current_stream_capturing() else None
        for i in range(1, opt.n_past + opt.n_future):
            h, skip = self._skip_rule(i, self._enc(enc_all, x, i - 1), skip)
            h_target = self._enc(enc_all, x, i)[0]
            if side is not None:
                side.wait_stream(cur)          # this step's encodings (already there when the frames were encoded up front)
            with torch.cuda.stream(side if side is not None else cur):
                h_pred = self.frame_predictor(h)
                mse_latent = mse_latent + self.mse_latent_criterion(h_pred, h_target)
                gp_pred = self.gp_layer(self._gp_in(h))
                max_ll = max_ll - self.mll(gp_pred, h_target.transpose(0, 1))
                gp_mean = gp_pred.mean.transpose(0, 1)
            if side is not None:
                cur.wait_event(side.record_event())
                # ORDERING INVARIANT: everything the main stream consumes from the side stream was allocated BY the side
                # stream; the caching allocator only knows the allocating stream, so tell it about the consumer - a block
                # freed (in backward: by the main stream's nodes) must not be handed out again on the side stream while
                # main-stream kernels still read it.  (Until r03 safety rested on the wait_stream at the top of each step.)
                for t_ in (h_pred, gp_mean):
                    t_.record_stream(cur)
            with self._share_scope():   # the three decoder calls of a step share the skip halves of their concat convs
                x_pred = self.decoder([h_pred, skip])
                x_target_pred = self.decoder([h_target, skip])
                x_pred_gp = self.decoder([gp_mean, skip])
            ae_mse = ae_mse + self.mse_latent_criterion(x_target_pred, x[i])
            mse = mse + self.mse_criterion(x_pred, x[i])
            mse_gp = mse_gp + self.mse_latent_criterion(x_pred_gp, x[i])
        if side is not None:
            cur.wait_stream(side)
        loss = 1000 * ae_mse + 0.001 * mse + 0.01 * mse_latent + 0.001 * mse_gp + 0.0001 * max_ll.sum()
        loss.backward()
        if side is not None:
            cur.wait_stream(side)              # the latent path's backward kernels
        if staged:
            # gradients of GP, likelihood, LSTM and decoder are final: their all-reduce runs under the encoder phase
            self._ar(("start", "a", (self.rng_gp[0], self.rng_dec[1])))
            outs, seeds = [], []
            for (h, sk), (hd, skd) in zip(enc_out, enc_all):
                for t, d in [(h, hd)] + list(zip(sk, skd)):
                    if d.grad is not None:
                        outs.append(t)
                        seeds.append(d.grad)
            torch.autograd.backward(outs, seeds)
            self._ar(("start", "b", self.rng_enc), ("finish", "a"), ("finish", "b"))
        else:
            self._ar(("reduce", (self.rng_gp[0], self.rng_enc[1])))
        self.frame_predictor_optimizer.step()
        self.encoder_optimizer.step()
        self.decoder_optimizer.step()
        self.optimizer.step()
        return mse_latent.detach(), loss.detach()

    def _finetune_dev(self, x):
        """Both fine-tuning closures of an iteration (train.py:175-198 then :146-172).  They are independent of each other: the
        LSTM closure reads the encoder and the LSTM and steps only the LSTM, the GP closure reads the encoder and the GP and
        steps only the GP, and neither steps the encoder.  So both backward passes run first, then ONE all-reduce over the
        adjacent arena ranges [GP | likelihood | LSTM] (r06: was one collective - and one cut of the hipGraph chain - per
        closure), then both Adam steps: the same arithmetic as the reference's order, three collectives and four graph
        segments per data-parallel iteration instead of four and five."""
        fp = self._train_fp_dev(x, defer_step=True)
        gp = self._train_gp_dev(x, defer_step=True)
        self._ar(("reduce", (self.rng_gp[0], self.rng_fp[1])))
        self.frame_predictor_optimizer.step()
        self.optimizer.step()
        return fp, gp

    def finetune_temporal_encoders(self, x):
        fp, gp = self._finetune_dev(x)
        return (float(fp) + float(gp)) / (self.opt.n_past + self.opt.n_future)

    def optimizers(self):
        return [self.frame_predictor_optimizer, self.encoder_optimizer, self.decoder_optimizer, self.optimizer]

    def iteration(self, x):
        """One iteration of the training loop (train.py:354-361): train_model, then the two fine-tuning closures when
        opt.ft.  Returns (mse_ctrl, indices, temp_loss) as Python floats."""
        mse_ctrl, indices = self.train_model(x)
        temp_loss = self.finetune_temporal_encoders(x) if self.opt.ft else 0
        self._iters = getattr(self, '_iters', 0) + 1
        return mse_ctrl, indices, temp_loss


    # ---- qualitative rollout of train.py:256-289 (tensors only; PNG/GIF writers are out of scope) ------
    @torch.no_grad()
    def plot(self, x, epoch, nsample=5, eps_by_sample=None):
        """train.py:256-310 without the image writers: `nsample` rollouts whose ONE GP-sampled step is i == 10 (:281;
        the GP is fed the encoder output h, :283), then per batch row the sample with the smallest summed squared error
        (:303-310; ties keep the first sample, like the reference's strict `<`).  `eps_by_sample[s]`: base sample (D,B)
        of rollout s (parity runs); None = torch RNG.  Returns (gen (S,T,B,C,H,W), best (B,))."""
        opt = self.opt
        gen_seq = []
        for s in range(nsample):
            self.frame_predictor.hidden = self.frame_predictor.init_hidden()
            seq = [x[0]]
            x_in = x[0]
            skip = None
            for i in range(1, opt.n_eval):
                h, skip = self._skip_rule(i, self.encoder(x_in), skip)
                if i < opt.n_past:
                    if self.encoder.training:
                        # train.py:273-274 encodes x[i] and discards it; with the encoder left in train mode (:372-374) that
                        # call still updates the BatchNorm running statistics the checkpoint will carry
                        self.encoder(x[i])
                    self.frame_predictor(h)
                    x_in = x[i]
                else:
                    h_pred = self.frame_predictor(h)
                    if i == 10:  # train.py:281: the one GP-sampled step of the qualitative rollout
                        pred = self.likelihood(self.gp_layer(self._gp_in(h)))
                        z = pred.rsample(None if eps_by_sample is None else eps_by_sample[s])
                        x_in = self.decoder([z.transpose(0, 1), skip])
                    else:
                        x_in = self.decoder([h_pred, skip])
                seq.append(x_in)
            gen_seq.append(torch.stack(seq))
        gen = torch.stack(gen_seq)                       # (S,T,B,C,H,W)
        gt = torch.stack(list(x[:opt.n_eval]))           # (T,B,C,H,W)
        sse = ((gen - gt.unsqueeze(0)) ** 2).sum((1, 3, 4, 5))   # (S,B) — train.py:303-310 best-of-N
        return gen, sse.argmin(0)

    def save(self, path):
        """train.py:380-388: whole-module pickles + GP / likelihood / GP-optimiser state dicts.  Every tensor written owns
        its storage: the live parameters, gradients and Adam moments are views of the shared FlatArena, and torch.save
        writes the whole storage behind a view (a checkpoint would carry the arena instead of the GP's moments, and loading
        only ck['encoder'] would pin all of it)."""
        def own(sd):
            out = type(sd)((k, v.detach().clone() if torch.is_tensor(v) else v) for k, v in sd.items())
            if hasattr(sd, "_metadata"):          # module version info torch's state_dict carries (and load_state_dict reads)
                out._metadata = sd._metadata
            return out
        torch.save({'encoder': _detached_copy(self.encoder), 'decoder': _detached_copy(self.decoder),
                    'frame_predictor': _detached_copy(self.frame_predictor),
                    'likelihood': own(self.likelihood.state_dict()), 'gp_layer': own(self.gp_layer.state_dict()),
                    'gp_layer_optimizer': self.optimizer.state_dict(), 'opt': self.opt}, path)


def _detached_copy(module):
    """A deep copy of `module` whose parameters and buffers are fresh tensors with their OWN storage and no gradient
    (copy.deepcopy alone clones the whole storage behind every arena view, and would copy `.grad` as well)."""
    import copy
    memo = {}
    for p in module.parameters():
        memo[id(p)] = torch.nn.Parameter(p.detach().clone(), requires_grad=p.requires_grad)
    for b in module.buffers():
        memo[id(b)] = b.detach().clone()
    hidden = getattr(module, "hidden", None)
    if hidden is not None:         # lstm.hidden: the recurrent state of the last sequence, may carry an autograd graph
        memo[id(hidden)] = [(h.detach().clone(), c.detach().clone()) for h, c in hidden]
    return copy.deepcopy(module, memo)


# hipGraph replay of the iteration and the batch prefetcher: dvg_amd/train_graphs.py (r06; re-exported for `train.<name>`)
from dvg_amd.train_graphs import BatchPrefetcher, GraphedIteration, SegmentedIteration  # noqa: E402,F401


def main(argv=None):
    opt = build_parser().parse_args(argv)
    opt.ft = not opt.no_ft
    if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
        # the latent path of an eager iteration runs on a second stream on purpose (autograd.JOIN_STREAMS joins them)
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    rank, world, local = parallel.init_distributed()
    opt.rank, opt.world = rank, world
    opt.local_batch = parallel.shard_batch(opt.batch_size, world)
    if rank == 0:
        print("Random Seed: ", opt.seed)
    random.seed(opt.seed + rank)
    np.random.seed(opt.seed + rank)
    torch.manual_seed(opt.seed)          # identical init on every rank (also broadcast below)
    assert torch.cuda.is_available(), "train.py needs a GPU: the DVG hot path has no CPU fallback"
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    torch.cuda.manual_seed_all(opt.seed)
    if rank == 0:
        print(opt)
    tr = Trainer(opt, device)
    torch.manual_seed(opt.seed + 1000 * rank)   # from here on: per-rank randomness (GP samples)
    train_gen = BatchPrefetcher(make_batch_generator(opt, opt.n_past + opt.n_future, opt.seed + 17 * rank, device))
    test_gen = make_batch_generator(opt, opt.n_eval, opt.seed + 7919 + 17 * rank, device)
    # One rank: the iteration as one hipGraph.  Several ranks: a chain of hipGraphs cut at the gradient all-reduces, which
    # stay eager (SegmentedIteration).  Capturing the RCCL collectives inside ONE graph instead is NOT safe
    # on this stack: the c10d watchdog thread may query a collective's event while it is "recorded in a capturing stream"
    # (hipErrorCapturedEvent) and terminate the process (1 of 5 runs with a one-rank RCCL group).
    if opt.no_hip_graph or tr.sync_bn:
        step = tr.iteration
    elif world == 1:
        step = GraphedIteration(tr)
    else:
        step = SegmentedIteration(tr)
    for epoch in range(opt.niter):
        tr.train_mode()
        tr.scheduler.step()   # before the epoch, as train.py:347
        epoch_mse = 0.0
        t0 = time.time()
        indices = 0.0
        for i in range(opt.epoch_size):
            x = next(train_gen)()
            mse_ctrl, indices, temp_loss = step(x)
            epoch_mse += mse_ctrl + temp_loss
        torch.cuda.synchronize()
        if rank == 0:
            fps = opt.batch_size * (opt.n_past + opt.n_future - 1) * opt.epoch_size / (time.time() - t0)
            print('[%02d] mse loss: %.5f (%d) %.5f' % (epoch, epoch_mse / opt.epoch_size,
                                                       epoch * opt.epoch_size * opt.batch_size, indices))
            print('     train frames/s: %.1f' % fps)
        if epoch % opt.save_every == 0:
            tr.frame_predictor.eval()
            tr.gp_layer.eval()
            tr.likelihood.eval()   # encoder / decoder stay in train mode, as train.py:372-374
            test_x = next(test_gen)()
            gen, best = tr.plot(test_x, epoch)
            if rank == 0 and not opt.no_save:
                os.makedirs(opt.output_path, exist_ok=True)
                torch.save({'gen': gen[:, :, :min(opt.local_batch, 10)].cpu(), 'best': best.cpu()},
                           '%s/sample_%d.pt' % (opt.output_path, epoch))
                tr.save('%s/model.pth' % opt.output_path)
        if epoch % 10 == 0 and rank == 0:
            print('log dir: %s' % opt.log_dir)
    if opt.print_param_checksum:   # tests: every rank must end with the same parameters
        mods = (tr.encoder, tr.decoder, tr.frame_predictor, tr.gp_layer, tr.likelihood)
        cs = sum(float(p.detach().double().sum()) for m in mods for p in m.parameters())
        ab = sum(float(p.detach().double().abs().sum()) for m in mods for p in m.parameters())
        print('rank %d param checksum %.17g %.17g' % (rank, cs, ab), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return tr


if __name__ == '__main__':
    main()
