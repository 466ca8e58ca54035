#!/usr/bin/env python3
"""DIAGNOSTIC (r06, profiles/r06_dp_race_bisect.txt: how the rehearsal's run-to-run different vgg_64 gradients were traced to the
F(4x4) weight transform).  The FIRST train_model backward, repeated `--repeats` times INSIDE one launch on the same batch from the
same parameters (optimiser steps disabled); every repeat's encoder / decoder / LSTM gradients are compared bit for bit with the
first, and a checksum of the last repeat is printed for launch-to-launch comparison.  Runs as one process or under
torch.distributed.run (DVG_DP_SHARE_GPU=1 DVG_DP_BACKEND=gloo: the one-GPU rehearsal; --batch is the PER-RANK batch).
  --meet sync | gloo   what happens at every BatchNorm call (stream synchronise; + a host-side all-reduce: ranks in lock-step)
  --noise mm | nan     a second STREAM of the process kept busy (NaN operands: poisoned LDS / register leftovers)
  --noise proc | procnan | proctrain   a second PROCESS on the device (matmuls / another trainer)
  --drop_caches        forget the per-weight-version caches before every repeat (the amplifier that exposed the fault)

  python tools/diag_repeat_backward.py --model vgg --batch 4 --repeats 40 --drop_caches --noise proctrain
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="vgg")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--n_past", type=int, default=2)
    ap.add_argument("--n_future", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=30)
    ap.add_argument("--meet", default="none", choices=["none", "sync", "sleep", "gloo"],
                    help="what happens at every BatchNorm call, forward and backward: sync = stream synchronise; gloo (under "
                         "torch.distributed.run, DVG_DP_SHARE_GPU=1 DVG_DP_BACKEND=gloo) = synchronise + a host-side all-reduce of "
                         "one float: the ranks resume at the same instant, as under --sync_bn")
    ap.add_argument("--drop_caches", action="store_true", help="forget the per-weight-version caches (packed / Winograd-domain / "
                    "transposed weights) before every repeat: a launch whose FIRST pass built a bad cache entry then shows it")
    ap.add_argument("--no_allreduce", action="store_true", help="ranks > 1: gradient all-reduce off (every rank keeps its own gradients)")
    ap.add_argument("--diag", default="", help="comma list: nolatent, wgrad1, winoffN / wino2_N (3x3 layers on N x N maps direct / F(2x2))")
    ap.add_argument("--noise", default="none", choices=["none", "mm", "nan", "proc", "procnan", "proctrain"],
                    help="a background thread keeps ANOTHER stream of this process busy with 2048^2 fp32 matmuls (nan: of NaN-filled "
                         "operands, so that whatever those waves leave behind in LDS / registers is poison) - the contention a "
                         "second process on the device causes, without the second process")
    ap.add_argument("--noise_child", default="", help="(internal) run as the noise process of --noise proc / procnan / proctrain")
    ap.add_argument("--fresh_batches", action="store_true", help="a new batch per repeat pair (two repeats per batch)")
    a = ap.parse_args()
    import time
    import torch
    import train
    import utils
    from dvg_amd import fused, ops
    from dvg_amd.data import SyntheticMovingMNIST
    if hasattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch"):
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
    from dvg_amd import parallel
    rank, world, local = parallel.init_distributed() if "RANK" in os.environ else (0, 1, 0)
    dev = torch.device("cuda", local)
    torch.cuda.set_device(local)
    if a.noise_child in ("mm", "nan"):
        # the noise of --noise proc / procnan: ANOTHER PROCESS on the device (own queues, own VMID), until the parent ends it
        fillv = float("nan") if a.noise_child == "nan" else 1.0
        na = torch.full((2048, 2048), fillv, device=dev)
        nc = torch.empty((2048, 2048), device=dev)
        torch.mm(na, na, out=nc)
        torch.cuda.synchronize()
        print("ready", flush=True)
        t_end, ppid0 = time.time() + 280, os.getppid()
        while time.time() < t_end and os.getppid() == ppid0:
            for _ in range(4):
                torch.mm(na, na, out=nc)
                nc.add_(na)
            torch.cuda.synchronize()
        return
    child = None
    if a.noise.startswith("proc"):
        import subprocess
        kind = {"proc": "mm", "procnan": "nan", "proctrain": "train"}[a.noise]
        cmd = [sys.executable, os.path.abspath(__file__), "--noise_child", kind, "--model", a.model, "--batch", str(a.batch),
               "--repeats", "100000", "--meet", a.meet]
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
        line = child.stdout.readline()
        assert line.strip() == "ready", line
    argv = ["--model", a.model, "--dataset", "smmnist", "--batch_size", str(a.batch), "--n_past", str(a.n_past),
            "--n_future", str(a.n_future), "--no_save"]
    opt = train.build_parser().parse_args(argv)
    opt.ft, opt.rank, opt.world = True, rank, world
    opt.local_batch = a.batch          # --batch is the PER-RANK batch here
    torch.manual_seed(5)
    tr = train.Trainer(opt, dev)
    tr.train_mode()
    if a.no_allreduce:
        tr.set_allreduce(False)
    diag = set(filter(None, a.diag.split(",")))
    if "nolatent" in diag:
        tr.latent_stream = False
    if "wgrad1" in diag:
        from dvg_amd import autograd as ag
        ag.WGRAD_BATCH, ag.DENSE_BATCH = 1, 1
    for d_ in diag:
        if d_.startswith("winoff") or d_.startswith("wino2_"):
            h_ = int(d_.replace("winoff", "").replace("wino2_", ""))
            for c_ in (64, 128, 256, 512, 1024):
                for co_ in (64, 128, 256, 512):
                    fused.WINOGRAD_LAYER_OVERRIDE[(c_, h_, co_)] = 0 if d_.startswith("winoff") else 2
    if a.meet != "none":
        tb1, bwd2 = fused._train_bn, ops.bn_act_bwd

        token = torch.zeros(1)

        def meet():
            if a.meet == "sync":
                torch.cuda.current_stream().synchronize()
            elif a.meet == "gloo":
                torch.cuda.current_stream().synchronize()
                if world > 1:
                    torch.distributed.all_reduce(token)
            else:
                time.sleep(0.001)

        def tb_meet(*ar, **kw):
            meet()
            return tb1(*ar, **kw)

        def bwd_meet(*ar, **kw):
            meet()
            return bwd2(*ar, **kw)
        fused._train_bn, ops.bn_act_bwd = tb_meet, bwd_meet
    T = a.n_past + a.n_future
    gen = SyntheticMovingMNIST(seq_len=T, seed=77)
    mods = {"encoder": tr.encoder, "decoder": tr.decoder, "frame_predictor": tr.frame_predictor, "gp_layer": tr.gp_layer,
            "likelihood": tr.likelihood}

    def grads():
        out = {}
        for name, m in mods.items():
            for k, p in m.named_parameters():
                if p.grad is not None:
                    out[f"{name}.{k}"] = p.grad.detach().clone()
        return out
    for o in tr.optimizers():
        o.step = lambda *args, **kw: None
    stop = None
    if a.noise in ("mm", "nan"):
        import threading
        stop = threading.Event()
        ns = torch.cuda.Stream()
        fillv = float("nan") if a.noise == "nan" else 1.0
        na = torch.full((2048, 2048), fillv, device=dev)
        nb = torch.full((2048, 2048), fillv, device=dev)
        nc = torch.empty((2048, 2048), device=dev)

        def noise():
            torch.cuda.set_device(0)
            with torch.cuda.stream(ns):
                while not stop.is_set():
                    for _ in range(4):
                        torch.mm(na, nb, out=nc)
                        nc.add_(na)
                    ns.synchronize()
        th = threading.Thread(target=noise, daemon=True)
        th.start()
    x = None
    first, differ, worst = None, 0, {}
    ppid0, t_child_end = os.getppid(), time.time() + 280
    for r in range(a.repeats):
        if a.noise_child and (time.time() > t_child_end or os.getppid() != ppid0):
            break               # a noise child never outlives its parent (or 280 s)
        if x is None or (a.fresh_batches and r % 2 == 0):
            xg, _ = utils.normalize_data(opt, torch.cuda.FloatTensor, gen.batch(a.batch * world))
            x = [t[rank * a.batch:(rank + 1) * a.batch].contiguous() for t in xg]
            first = None
        if a.noise_child == "train" and r == 1:
            print("ready", flush=True)
        if a.drop_caches:
            from dvg_amd import autograd as ag_
            ag_._pack_cache.clear()
        tr.optimizer.zero_grad()      # the GP / likelihood gradients, which train_model leaves to accumulate (reference behaviour)
        tr._train_model_dev(x)
        torch.cuda.current_stream().synchronize()
        g = grads()
        if first is None:
            first = g
            continue
        bad = {}
        for k, t in g.items():
            if not torch.equal(t, first[k]):
                bad[k] = float((t - first[k]).abs().max() / first[k].abs().max().clamp_min(1e-30))
        if bad:
            differ += 1
            top = sorted(bad.items(), key=lambda kv: -kv[1])[:4]
            print(f"[rank {rank}] repeat {r}: {len(bad)} of {len(g)} tensors differ; worst {[(k, f'{v:.1e}') for k, v in top]}", flush=True)
            for k, v in bad.items():
                worst[k] = max(worst.get(k, 0.0), v)
    if stop is not None:
        stop.set()
        th.join()
    if child is not None:
        child.terminate()       # the exact process started above
        child.wait()
    print(f"diag_repeat_backward[rank {rank} of {world}, all-reduce {tr.reducer.active()}]: model {a.model} batch {a.batch} T {T} meet {a.meet} noise {a.noise} diag {sorted(diag)}: "
          f"{differ} of {a.repeats - 1} repeats differ from the first", flush=True)
    # launch-to-launch comparison: a checksum of the LAST repeat's gradients (exact: sums of |g| in fp64, repr'd)
    tot = sum(float(t.double().abs().sum()) for k, t in g.items() if not k.startswith(("gp_layer", "likelihood")))
    pick = [k for k in g if k.endswith(("upc3.0.main.0.weight", "c2.0.main.0.weight", "upc5.0.main.0.weight"))][:3]
    print(f"[rank {rank}] checksum {tot!r} " + " ".join(f"{k}={float(g[k].double().abs().sum())!r}" for k in pick), flush=True)
    if worst:
        print(f"[rank {rank}] tensors that ever differed:", sorted(worst.items(), key=lambda kv: -kv[1])[:12], flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
