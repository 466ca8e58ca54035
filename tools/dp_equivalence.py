#!/usr/bin/env python3
"""Data-parallel training == single-process training.

The reference is ONE process: BatchNorm statistics, the losses' means and the ELBO cover the whole batch (train.py:89-91,
200-248).  This worker runs `--iters` training iterations (train_model + both fine-tuning closures, eager) on ONE global batch
sequence that every rank generates identically and of which rank r takes clips [r B/N, (r+1) B/N); rank 0 writes, per
iteration, the loss values and - after the first backward / at the end - gradients, parameters and BatchNorm buffers to
`--out`.  tests/test_gpu_multirank.py compares a 2-rank `--sync_bn` run with the 1-rank run (must agree to fp32 rounding) and
with the 2-rank per-replica-BatchNorm run (must NOT: the control).

  python tools/dp_equivalence.py --model dcgan --batch 8 --iters 3 --out one.pt
  DVG_DP_SHARE_GPU=1 DVG_DP_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
      --master-port P tools/dp_equivalence.py --model dcgan --batch 8 --iters 3 --sync_bn --out two.pt
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="dcgan")
    ap.add_argument("--batch", type=int, default=8, help="GLOBAL batch")
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--n_past", type=int, default=2)
    ap.add_argument("--n_future", type=int, default=3)
    ap.add_argument("--sync_bn", action="store_true")
    ap.add_argument("--linear_lrelu", action="store_true",
                    help="DIAGNOSTIC: every train-mode LeakyReLU with slope 1 (forward and backward).  Two runs whose forward values "
                         "agree to rounding still take the odd LeakyReLU branch differently (a pre-activation within ~1e-7 of "
                         "zero), and the backward pass amplifies that seed layer by layer; without the kinks the comparison shows "
                         "what sync-BN itself leaves")
    ap.add_argument("--diag", default="", help="DIAGNOSTIC: comma list of sync_ar (device synchronise before every gradient "
                    "all-reduce action), wgrad1 (no deferred weight-gradient queues), nolatent (latent path on the main stream)")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import torch
    import train
    import utils
    from dvg_amd import parallel
    from dvg_amd.data import SyntheticMovingMNIST
    if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    if a.linear_lrelu:
        from dvg_amd import ops
        apply0, bwd0 = ops.bn_act_apply, ops.bn_act_bwd
        ops.bn_act_apply = lambda *ar, **kw: apply0(*ar, **dict(kw, slope=1.0))
        ops.bn_act_bwd = lambda *ar, **kw: bwd0(*ar, **dict(kw, slope=1.0))
    rank, world, local = parallel.init_distributed()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    argv = ["--model", a.model, "--dataset", "smmnist", "--batch_size", str(a.batch), "--n_past", str(a.n_past),
            "--n_future", str(a.n_future), "--no_save"] + (["--sync_bn"] if a.sync_bn else [])
    opt = train.build_parser().parse_args(argv)
    opt.ft, opt.rank, opt.world = True, rank, world
    opt.local_batch = parallel.shard_batch(a.batch, world)
    torch.manual_seed(5)
    tr = train.Trainer(opt, dev)
    tr.train_mode()
    diag = set(filter(None, a.diag.split(",")))
    if "sync_ar" in diag:
        run0 = tr._run_ar

        def run_ar(actions):
            torch.cuda.synchronize()
            run0(actions)
        tr._run_ar = run_ar
    if "wgrad1" in diag:
        from dvg_amd import autograd as ag
        ag.WGRAD_BATCH, ag.DENSE_BATCH = 1, 1
    if "nolatent" in diag:
        tr.latent_stream = False
    if tr.sync_bn and ("nofwd" in diag or "nobwd" in diag or "nocomm" in diag):
        # bisecting sync-BN: without its forward half / its backward half / its communication (the arithmetic is then not the
        # single-process one: only run-to-run determinism is being tested)
        from dvg_amd import fused, ops
        from dvg_amd.ops import norm
        if "nofwd" in diag:
            tb0 = fused._train_bn
            fused._train_bn = lambda bn, stats, count, save=False, synced=False: tb0(bn, stats, count, save=save, synced=True)
        if "nobwd" in diag:
            bwd1 = ops.bn_act_bwd

            def unsynced(*ar, **kw):
                keep, norm.SYNC_BN = norm.SYNC_BN, None
                try:
                    return bwd1(*ar, **kw)
                finally:
                    norm.SYNC_BN = keep
            ops.bn_act_bwd = unsynced
        if "nocomm" in diag:
            class _NoComm:
                @staticmethod
                def all_reduce(t, group=None):
                    t.mul_(2.0)

                @staticmethod
                def get_backend(group=None):
                    return "gloo"
            norm.SYNC_BN = (_NoComm, norm.SYNC_BN[1], norm.SYNC_BN[2])
    for d_ in diag:
        if d_.startswith("winoff") or d_.startswith("wino2_"):      # bisecting: 3x3 layers on h x h maps in direct / F(2x2) form
            from dvg_amd import fused
            h_ = int(d_.replace("winoff", "").replace("wino2_", ""))
            for c_ in (64, 128, 256, 512, 1024):
                for co_ in (64, 128, 256, 512):
                    fused.WINOGRAD_LAYER_OVERRIDE[(c_, h_, co_)] = 0 if d_.startswith("winoff") else 2
    if "dgrad_direct16" in diag or "fwd_direct16" in diag:
        from dvg_amd import autograd as ag, fused
        keys16 = {(c_, 16, co_): 0 for c_ in (64, 128, 256, 512, 1024) for co_ in (64, 128, 256, 512)}

        def with_direct16(fn):
            def inner(*ar, **kw):
                fused.WINOGRAD_LAYER_OVERRIDE.update(keys16)
                try:
                    return fn(*ar, **kw)
                finally:
                    for k_ in keys16:
                        fused.WINOGRAD_LAYER_OVERRIDE.pop(k_, None)
            return inner
        if "dgrad_direct16" in diag:
            ag._dgrad3 = with_direct16(ag._dgrad3)
        if "fwd_direct16" in diag:
            ag._conv3_raw = with_direct16(ag._conv3_raw)
    if "lockstep" in diag and world > 1:
        # per-replica BatchNorm, but the ranks meet (a host-side all-reduce of one float) at every BatchNorm call, forward and
        # backward: the lock-step that sync-BN's collectives impose, without its arithmetic
        from dvg_amd import fused, ops
        token = torch.zeros(1)
        tb1, bwd2 = fused._train_bn, ops.bn_act_bwd

        def meet():
            torch.cuda.current_stream().synchronize()
            torch.distributed.all_reduce(token)

        def tb_meet(*ar, **kw):
            meet()
            return tb1(*ar, **kw)

        def bwd_meet(*ar, **kw):
            meet()
            return bwd2(*ar, **kw)
        fused._train_bn, ops.bn_act_bwd = tb_meet, bwd_meet
    if "sync_bwd" in diag or "sleep_bwd" in diag:
        # a host-side stall at every BatchNorm backward (what sync-BN's collective does to the timing), without any collective
        import time
        from dvg_amd import ops
        bwd0 = ops.bn_act_bwd

        def stalled(*ar, **kw):
            if "sync_bwd" in diag:
                torch.cuda.synchronize()
            else:
                time.sleep(0.002)
            return bwd0(*ar, **kw)
        ops.bn_act_bwd = stalled
    T = a.n_past + a.n_future
    gen = SyntheticMovingMNIST(seq_len=T, seed=77)        # the SAME generator state on every rank: the global batches
    lo, hi = rank * opt.local_batch, (rank + 1) * opt.local_batch
    rec = {"world": world, "sync_bn": bool(tr.sync_bn), "losses": []}
    mods = {"encoder": tr.encoder, "decoder": tr.decoder, "frame_predictor": tr.frame_predictor, "gp_layer": tr.gp_layer,
            "likelihood": tr.likelihood}

    def flat(kind):
        out = {}
        for name, m in mods.items():
            src = m.named_parameters() if kind != "buffers" else m.named_buffers()
            for k, p in src:
                t = p.grad if kind == "grads" else p
                if t is not None and t.is_floating_point():
                    out[f"{name}.{k}"] = t.detach().float().cpu().clone()
        return out
    for it in range(a.iters):
        xg, _ = utils.normalize_data(opt, torch.cuda.FloatTensor, gen.batch(a.batch))
        x = [t[lo:hi].contiguous() for t in xg]
        if it == 0:
            # the first train_model backward alone, WITHOUT its optimiser steps: the averaged gradients themselves
            steps = [o.step for o in tr.optimizers()]
            for o in tr.optimizers():
                o.step = lambda *args, **kw: None
            try:
                tr._train_model_dev(x)
            finally:
                for o, s in zip(tr.optimizers(), steps):
                    o.step = s
            torch.cuda.synchronize()
            rec["grads_first_backward"] = flat("grads")
            rec["buffers_first_forward"] = flat("buffers")
            continue                                  # (the BatchNorm buffers have advanced once on every rank alike)
        if it == 1:
            # the first iteration that steps, in its two halves (= Trainer.iteration): every run starts it from identical
            # parameters, so train_model's loss values and the parameters right after ITS Adam steps are directly comparable;
            # the fine-tuning closures then already run on stepped weights
            mse, _ = tr.train_model(x)
            torch.cuda.synchronize()
            rec["params_first_step"] = flat("params")
            temp = tr.finetune_temporal_encoders(x)
        else:
            mse, _, temp = tr.iteration(x)
        vals = torch.tensor([float(mse), float(tr.last_loss), float(temp)], dtype=torch.float64, device=dev)
        if world > 1:       # a rank's loss values are those of ITS clips: the mean over the ranks is the global batch's
            torch.distributed.all_reduce(vals)
            vals /= world
        rec["losses"].append(tuple(float(v) for v in vals))
    torch.cuda.synchronize()
    rec["params"] = flat("params")
    rec["buffers"] = flat("buffers")
    if rank == 0:
        torch.save(rec, a.out)
        print(f"dp_equivalence: world {world} sync_bn {rec['sync_bn']} losses {rec['losses']}", flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
