#!/usr/bin/env python3
"""generate_frames.py — counterpart of the reference's generate_frames.py on the MI355X kernel library.

Same flags (generate_frames.py:17-41).  Loads `<model_dir>/<dataset>.pth` (falls back to the
`model.pth` that train.py writes — the reference's two scripts disagree on the filename),
takes `opt` from the checkpoint (:44), forces n_eval / n_future / batch_size like :47-49 unless
overridden, and runs either

  * `make_gifs` (:107-217): the posterior rollout (GP fed the LSTM output, predictive MEAN decoded) plus
    `nsample` diverse rollouts where a GP sample replaces the LSTM prediction at steps with i % 15 == 0, or
  * `GPtrigger_gen` (:249-300, `--gp_trigger`): per batch index, a 12-step warm-up that records the norm of
    the GP predictive variance, then a variance-threshold trigger `value > mean + (2+0.01*depth)*std` over a
    12-long sliding window decides per step between a GP sample and the LSTM path.

Image/GIF writing and SSIM are out of scope: results (frames, per-sample PSNR, best-of-N
index, trigger steps) are saved as tensors.  `--synthetic_ckpt` builds a randomly initialised checkpoint so
the script can run without a trained model.
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import utils  # noqa: E402
from dvg_amd import ops  # noqa: E402
from dvg_amd.data import SyntheticMovingMNIST, synthetic_video  # noqa: E402
from dvg_amd.rollout import (GraphedSampler, GraphedTrigger, condition, posterior_from, sample_from, sample_rollout,  # noqa: E402
                             trigger_body, trigger_log, trigger_warmup)
from gp_models import GaussianLikelihood, GPRegressionLayer1  # noqa: E402


def build_parser():
    p = argparse.ArgumentParser()
    p.add_argument('--batch_size', default=50, type=int)
    p.add_argument('--log_dir', default='logs_gp')
    p.add_argument('--model_dir', default='')
    p.add_argument('--name', default='')
    p.add_argument('--data_root', default='./data/kth')
    p.add_argument('--seed', default=1, type=int)
    p.add_argument('--image_width', type=int, default=64)
    p.add_argument('--channels', default=1, type=int)
    p.add_argument('--gp_trigger', action='store_true', help='variance-threshold trigger (GPtrigger_gen)')
    p.add_argument('--dataset', default='kth')
    p.add_argument('--n_past', type=int, default=5)
    p.add_argument('--n_future', type=int, default=100)
    p.add_argument('--n_eval', type=int, default=105)
    p.add_argument('--rnn_size', type=int, default=256)
    p.add_argument('--predictor_rnn_layers', type=int, default=2)
    p.add_argument('--z_dim', type=int, default=10)
    p.add_argument('--g_dim', type=int, default=90)
    p.add_argument('--model', default='dcgan')
    p.add_argument('--data_threads', type=int, default=5)
    p.add_argument('--last_frame_skip', action='store_true')
    p.add_argument('--nsample', type=int, default=100)
    p.add_argument('--nbatches', type=int, default=5)
    p.add_argument('--trigger_indices', type=int, default=None, help='GPtrigger_gen: how many batch indices')
    p.add_argument('--synthetic_ckpt', action='store_true')
    p.add_argument('--inflight', type=int, default=3,
                   help='make_gifs: samples of a batch drawn at once (one hipGraph + stream each); 0 = eager loop, one at a time')
    p.add_argument('--no_share_prefix', action='store_true',
                   help='make_gifs: run the prediction steps before the first GP trigger step once per SAMPLE, like the reference '
                        'loop (default: once per batch - they are the same for every sample; identical results)')
    p.add_argument('--synthetic_data', action='store_true',
                   help='datasets other than smmnist: synthetic clips of the right shape (--data_root is not read)')
    return p


class Generator:
    def __init__(self, opt, ckpt, device):
        self.opt, self.dev = opt, device
        self.encoder, self.decoder = ckpt['encoder'], ckpt['decoder']
        self.frame_predictor = ckpt['frame_predictor']
        self.likelihood = GaussianLikelihood(batch_size=opt.g_dim)
        self.gp_layer = GPRegressionLayer1(opt.g_dim)
        self.likelihood.load_state_dict(ckpt['likelihood'])
        self.gp_layer.load_state_dict(ckpt['gp_layer'])
        for m in (self.encoder, self.decoder, self.frame_predictor, self.gp_layer, self.likelihood):
            m.to(device).eval()
        self.frame_predictor.batch_size = opt.batch_size
        self._sampler, self._sampler_key = None, None

    def _gp(self, h):
        return self.likelihood(self.gp_layer(h.transpose(0, 1).view(self.opt.g_dim, h.shape[0], 1)))

    @torch.no_grad()
    def make_gifs(self, x, nsample, eps_by_sample=None):
        """`eps_by_sample[s][i]`: the N(0,1) base sample (D,B) of sample s at trigger step i (parity runs); None = torch RNG."""
        opt = self.opt
        B, T = x[0].shape[0], opt.n_eval - opt.n_past
        ssim = torch.zeros(B, nsample, T, device=self.dev)
        psnr = torch.zeros(B, nsample, T, device=self.dev)
        all_gen = []
        inflight = getattr(opt, 'inflight', 3)
        if inflight > 0:
            # One graph per batch for everything the samples share (conditioning, the posterior rollout's prediction steps, the
            # samples' steps before the first GP trigger step) and one per chain for the sample body: rollout.GraphedSampler.
            # Parameter / buffer versions are part of the key: the captured graphs read packed weights and BatchNorm folds of
            # the versions they were captured with, so a weight change (load_state_dict, an optimiser step) re-captures
            share = not getattr(opt, 'no_share_prefix', False)

            def key():
                vers = tuple(t._version for m in (self.encoder, self.decoder, self.frame_predictor, self.gp_layer, self.likelihood)
                             for t in list(m.parameters()) + list(m.buffers()))
                return (tuple(x[0].shape), len(x), opt.n_past, opt.n_eval, bool(opt.last_frame_skip), inflight, share, vers)
            if self._sampler_key != key():
                self._sampler = GraphedSampler(self.encoder, self.decoder, self.frame_predictor, self.gp_layer,
                                               self.likelihood, x, opt.n_past, opt.n_eval, opt.last_frame_skip,
                                               inflight=inflight, share_prefix=None if share else False)
                # the key AFTER the construction: the sampler's eager warm-up pass may be the GP layer's first call, which
                # initialises its variational parameters in place (gp_models: variational_params_initialized); the graphs are
                # captured after that pass, i.e. with the versions read here
                self._sampler_key = key()
            self._sampler.set_batch(x)
            samples = torch.empty((nsample, opt.n_eval) + tuple(x[0].shape), device=self.dev)
            post = self._sampler.run(nsample, samples, ssim, psnr, eps_by_sample).clone()
            best = ssim.mean(2).argsort(1)[:, -1]
            return {'posterior': post, 'samples': samples, 'ssim': ssim, 'psnr': psnr, 'best': best}
        # everything before the first predicted frame is the same for the posterior rollout and all nsample rollouts of this
        # batch (generate_frames.py:113-121 and :147-162 are the same computation): once per batch
        state = condition(self.encoder, self.frame_predictor, x, opt.n_past, opt.last_frame_skip, decoder=self.decoder)
        post = posterior_from(state, self.encoder, self.decoder, self.frame_predictor, self.gp_layer, self.likelihood,
                              opt.n_past, opt.n_eval, opt.last_frame_skip)
        for s in range(nsample):
            frames = sample_from(state, self.encoder, self.decoder, self.frame_predictor, self.gp_layer,
                                 self.likelihood, opt.n_past, opt.n_eval, opt.last_frame_skip,
                                 eps_by_step=None if eps_by_sample is None else eps_by_sample[s])
            for t in range(T):   # utils.eval_seq (generate_frames.py:178) on device: dvg_eval_frames
                ssim[:, s, t], psnr[:, s, t] = ops.eval_frames(x[opt.n_past + t], frames[opt.n_past + t])
            all_gen.append(torch.stack(frames))
        best = ssim.mean(2).argsort(1)[:, -1]   # generate_frames.py:188-189,207: np.argsort(mean_ssim)[-1]
        return {'posterior': torch.stack(post), 'samples': torch.stack(all_gen), 'ssim': ssim, 'psnr': psnr,
                'best': best}

    @torch.no_grad()
    def _generation(self, x_in, skip):
        h = self.encoder(x_in)[0]
        return self.decoder([self.frame_predictor(h), skip])

    @staticmethod
    def _variance_norms(pred):
        """generate_frames.py:230,275: `np.linalg.norm(final_pred.variance.cpu().detach().numpy().transpose(), axis=1)` -
        per-sample L2 norm over the latent dims of the predictive variance, on the host in float32 like the reference."""
        return np.linalg.norm(pred.variance.cpu().numpy().transpose(), axis=1)

    @torch.no_grad()
    def gp_trigger_gen(self, x, n_index=None, warmup=12, total=105, depth=1, eps_by_step=None, keep_batch=False,
                       indices=None, graph=True, host_loop=False, share_paths=True):
        """generate_frames.py:249-298.  Keeps the reference's bookkeeping verbatim: the warm-up records the
        variance norm of sample `index` (:275) while `var_value` reads sample [3] (:230 - a batch smaller than 4 is an
        IndexError there and an error here); the skip tensors are those of loop steps `i < 5` (:268-269); the rollout is
        autoregressive from x[0]; a triggered step decodes a GP sample and does NOT step the LSTM (:289-292).
        `eps_by_step[i]`: base sample (D,B) for a trigger at step i (parity runs); None = torch RNG.
        indices: the batch indices to run (default range(n_index or B), the reference's `for index in range(batch_size)`).
        Default schedule (rollout.trigger_warmup / trigger_body): the warm-up once per BATCH, one encoder call per step,
        decision and branch select on the device, the main loop a hipGraph (`graph`), logs read back once per index;
        host_loop=True: the reference's own schedule (per index: warm-up, a host round trip and 2-3 encoder calls per step).
        share_paths: the main loop's value is that of sample [3] whatever the index (:230) - only the window's first entries (the
        warm-up norms of the index's own column, :275) differ between indices - so an index whose decisions on an ALREADY
        COMPUTED rollout's values (ops.gp_trigger_replay: the device decision's arithmetic) are that rollout's decisions IS that
        rollout: same frames, no launches.  Applied to rollouts WITHOUT a trigger only: a triggered step draws a GP sample
        from torch's generator, and the reference draws afresh for every index."""
        B = x[0].shape[0]
        if B < 4:
            raise IndexError("GPtrigger_gen reads sample [3] of the batch (generate_frames.py:230): batch_size must be >= 4")
        if indices is None:
            indices = range(B if n_index is None else n_index)
        if host_loop:
            return self._gp_trigger_gen_host(x, indices, warmup, total, depth, eps_by_step, keep_batch)
        mods = (self.encoder, self.decoder, self.frame_predictor, self.gp_layer, self.likelihood)
        out = []
        if graph:
            def key():     # parameter / buffer versions: the graphs read packed weights and folds of the versions they captured
                return (tuple(x[0].shape), warmup, total, depth,
                        tuple(t._version for m in mods for t in list(m.parameters()) + list(m.buffers())))
            if getattr(self, "_trigger_key", None) != key():
                self._trigger = GraphedTrigger(*mods, x[0], warmup, total, depth)
                self._trigger_key = key()      # AFTER the construction: its eager pass may initialise the GP in place (make_gifs)
            self._trigger.warm(x[0])
            run = lambda index: self._trigger.run(index, eps_by_step)   # noqa: E731
        else:
            dev = x[0].device
            state = trigger_warmup(*mods, x[0], warmup)
            warm_stack = torch.stack(state["frames"])
            eps = torch.zeros(max(1, total - warmup), self.opt.g_dim, B, device=dev)

            def run(index):
                log, ctx = trigger_log(total, dev), state["norms"][:, index].clone()
                if eps_by_step is None:
                    eps.normal_()
                else:
                    for i in range(warmup, total):
                        eps[i - warmup].copy_(eps_by_step[i])
                fr = trigger_body(state, *mods, ctx, 2 + 0.01 * depth, eps, log, warmup, total)
                flags = log["flags"][warmup:].cpu()
                return {"frames": torch.cat([warm_stack, torch.stack(fr)]) if fr else warm_stack,
                        "triggers": [warmup + int(i) for i in torch.nonzero(flags).flatten()],
                        "values": [float(v) for v in torch.cat([state["norms"][:, index], log["values"][warmup:]]).cpu()],
                        "thresholds": [float(v) for v in log["thresholds"][warmup:].cpu()],
                        "values_main": log["values"][warmup:].clone()}
        norms = (self._trigger.state if graph else state)["norms"]
        quiet = []          # computed rollouts without a trigger: (frames (clone), main-loop values on the device)
        self.trigger_rollouts_run = 0
        for index in indices:
            r = None
            if share_paths and total > warmup:
                for fr_q, vm_q in quiet:
                    fl, th = ops.gp_trigger_replay(vm_q, norms[:, index], 2 + 0.01 * depth)
                    if not bool(fl.any()):            # this index decides "no trigger" at every step of that rollout too
                        r = {"frames": fr_q, "triggers": [], "thresholds": [float(v) for v in th.cpu()],
                             "values": [float(v) for v in torch.cat([norms[:, index], vm_q]).cpu()]}
                        break
            if r is None:
                r = run(index)
                self.trigger_rollouts_run += 1
                if share_paths and not r["triggers"] and total > warmup:
                    quiet.append((r["frames"].clone(), r["values_main"]))
            out.append({'index': index, 'frames': r["frames"][:, index].cpu(), 'triggers': r["triggers"],
                        'values': r["values"], 'thresholds': r["thresholds"]})
            if keep_batch:        # parity tests compare the whole batch's frames, not only row `index`
                out[-1]['batch_frames'] = [t.clone() for t in r["frames"]]
        return out

    @torch.no_grad()
    def _gp_trigger_gen_host(self, x, indices, warmup, total, depth, eps_by_step, keep_batch):
        """The reference's schedule, statement for statement (see gp_trigger_gen): kept as the timing baseline of the device
        schedule and as its cross-check (tests/test_gpu_rollouts.py)."""
        out = []
        for index in indices:
            self.frame_predictor.hidden = self.frame_predictor.init_hidden()
            ctx, triggers, gen_seq, values, thresholds = [], [], [], [], []
            x_in, skip = x[0], None
            for i in range(warmup):
                h, sk = self.encoder(x_in)
                if i < 5:
                    skip = sk
                value = self._variance_norms(self._gp(h))[index]
                ctx.append(value)
                values.append(float(value))
                x_in = self._generation(x_in, skip)
                gen_seq.append(x_in)
            ctx = np.array(ctx)
            for i in range(warmup, total):
                h = self.encoder(x_in)[0]
                pred = self._gp(h)
                value = self._variance_norms(pred)[3]
                ctx = np.concatenate([ctx[1:], [value]])
                threshold = np.mean(ctx) + (2 + 0.01 * depth) * np.std(ctx)
                if value > threshold:
                    x_in = self.decoder([pred.rsample(None if eps_by_step is None else eps_by_step[i]).transpose(0, 1),
                                         skip])
                    triggers.append(i)
                else:
                    x_in = self._generation(x_in, skip)
                values.append(float(value))
                thresholds.append(float(threshold))
                gen_seq.append(x_in)
            out.append({'index': index, 'frames': torch.stack(gen_seq)[:, index].cpu(), 'triggers': triggers,
                        'values': values, 'thresholds': thresholds})
            if keep_batch:
                out[-1]['batch_frames'] = gen_seq
        return out


def synthetic_checkpoint(opt):
    import importlib
    import models.lstm as lstm_models
    model = importlib.import_module(f"models.{opt.model}_{opt.image_width}")
    enc, dec = model.encoder(opt.g_dim, opt.channels), model.decoder(opt.g_dim, opt.channels)
    enc.apply(utils.init_weights), dec.apply(utils.init_weights)
    fp = lstm_models.lstm(opt.g_dim, opt.g_dim, opt.rnn_size, opt.predictor_rnn_layers, opt.batch_size)
    fp.apply(utils.init_weights)
    gp, lik = GPRegressionLayer1(opt.g_dim), GaussianLikelihood(batch_size=opt.g_dim)
    return {'encoder': enc, 'decoder': dec, 'frame_predictor': fp, 'likelihood': lik.state_dict(),
            'gp_layer': gp.state_dict(), 'opt': opt}


def main(argv=None):
    args = build_parser().parse_args(argv)
    assert torch.cuda.is_available(), "generate_frames.py needs a GPU: the DVG hot path has no CPU fallback"
    device = torch.device('cuda', 0)
    if args.synthetic_ckpt:
        ckpt = synthetic_checkpoint(args)
        opt = args
    else:
        path = '%s/%s.pth' % (args.model_dir, args.dataset)
        if not os.path.exists(path):
            path = '%s/model.pth' % args.model_dir
        ckpt = torch.load(path, map_location='cpu', weights_only=False)
        opt = ckpt['opt']
        opt.n_eval, opt.n_future, opt.batch_size = args.n_eval, args.n_future, args.batch_size
        opt.log_dir = args.log_dir
        opt.inflight = args.inflight
    os.makedirs('%s/gen/' % opt.log_dir, exist_ok=True)
    print("Random Seed: ", opt.seed)
    random.seed(opt.seed)
    torch.manual_seed(opt.seed)
    torch.cuda.manual_seed_all(opt.seed)
    gen = Generator(opt, ckpt, device)
    dataset = getattr(opt, 'dataset', 'smmnist')
    if dataset != 'smmnist' and not args.synthetic_data:
        raise SystemExit(f"generate_frames.py: no loader for dataset {dataset} here (--data_root {args.data_root!r} is not "
                         "read). Pass --synthetic_data to roll out on synthetic clips of that dataset's shape.")
    print("WARNING: synthetic data - %s; --data_root is ignored" %
          ("Moving-MNIST trajectories over synthetic sprites (not MNIST digits)" if dataset == 'smmnist'
           else f"random textured clips shaped like {dataset}"), file=sys.stderr)
    if dataset == 'smmnist':
        ds = SyntheticMovingMNIST(seq_len=opt.n_eval, image_size=opt.image_width, seed=opt.seed + 7919)
        batches = (ds.batch(opt.batch_size) for _ in range(args.nbatches))
    else:
        batches = (synthetic_video(opt.batch_size, opt.n_eval, opt.channels, opt.image_width, seed=opt.seed + k)
                   for k in range(args.nbatches))
    for i, seq in enumerate(batches):
        test_x, _ = utils.normalize_data(opt, torch.cuda.FloatTensor, seq)
        if args.gp_trigger:
            res = gen.gp_trigger_gen(test_x, args.trigger_indices, total=opt.n_eval)
            torch.save(res, '%s/gen/gp_trigger_%d.pt' % (opt.log_dir, i))
            print('batch %d: trigger steps of index 0: %s' % (i, res[0]['triggers']))
        else:
            res = gen.make_gifs(test_x, args.nsample)
            torch.save({'posterior': res['posterior'][:, 0].cpu(), 'best': res['best'].cpu(), 'psnr': res['psnr'].cpu(),
                        'ssim': res['ssim'].cpu(),
                        'best_sample_0': res['samples'][int(res['best'][0]), :, 0].cpu()},
                       '%s/gen/sample_lstm_%d.pt' % (opt.log_dir, i))
            sel = res['best'].view(-1, 1, 1).expand(-1, 1, res['ssim'].shape[2])
            print('batch %d: best-of-%d by mean SSIM: SSIM %.4f, PSNR %.3f dB' % (
                i, args.nsample, float(res['ssim'].gather(1, sel).mean()), float(res['psnr'].gather(1, sel).mean())))


if __name__ == '__main__':
    main()
