#!/usr/bin/env python3
"""bench.py — predicted frames/s of the DVG inference rollout on MI355X.

A "step" is ONE rollout sample (one pass of the make_gifs sample loop, generate_frames.py:143-177)
over one batch of synthetic Moving-MNIST: B=64, 64x64, 10 conditioning + 10 predicted frames
(19 encoder calls, 10 decoder calls, 19 LSTM steps, 1 GP trigger sample at i=15), i.e.
BASELINE.json configs[1].  value = B * n_future * steps * n_gpus / time.

  python bench.py --gpus N --steps K --warmup W [--model vgg|dcgan]

For N>1 the driver launches one rank per GPU with torch.distributed.run; the rollout shards by
replicas (every GPU rolls out its own batch: no data-path collective), scaling = "weak".
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0


def build_models(model: str, batch: int, nc: int, dev, seed: int):
    import importlib
    from dvg_amd import utils
    from dvg_amd.models.gp_models import GaussianLikelihood, GPRegressionLayer1
    from dvg_amd.models.lstm import lstm
    torch.manual_seed(seed)
    m = importlib.import_module(f"dvg_amd.models.{model}_64")
    enc, dec = m.encoder(90, nc), m.decoder(90, nc)
    enc.apply(utils.init_weights)
    dec.apply(utils.init_weights)
    fp = lstm(90, 90, 256, 2, batch)
    fp.apply(utils.init_weights)
    gp, lik = GPRegressionLayer1(90), GaussianLikelihood(batch_size=90)
    mods = [enc, dec, fp, gp, lik]
    for x in mods:
        x.to(dev).eval()
    return mods


@torch.no_grad()
def calibrate_batchnorm(enc, dec, frame):
    """Give the BatchNorm layers the running statistics a trained model would have (batch statistics
    of the synthetic data, momentum 1) so that eval-mode activations stay O(1) through all layers
    instead of collapsing / exploding with the N(0,0.02) init (degenerate operands flatter DVFS)."""
    bns = [m for m in list(enc.modules()) + list(dec.modules()) if isinstance(m, torch.nn.BatchNorm2d)]
    for m in bns:
        m.momentum = 1.0
    enc.train(), dec.train()
    h, skips = enc(frame)
    dec([h, skips])
    for m in bns:
        m.momentum = 0.1
    enc.eval(), dec.eval()


def usable_cores() -> int:
    """Cores this process may actually use: min(affinity, cgroup CPU quota).  The GPU boxes show 256
    logical CPUs but run the job under a 16-CPU cgroup quota; 256 threads on 16 CPUs thrash."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(model: str, batch: int, n_past: int, n_eval: int, seed: int):
    """The oracle (CPU restatement, parity-checked against the reference's modules) timed on the host
    cores of this box: ONE full rollout of the same workload (bounded: ~10-30 s of CPU work)."""
    import importlib
    from oracle import dvg_oracle as orc
    from oracle import params
    from dvg_amd.data import SyntheticMovingMNIST
    cores = usable_cores()
    torch.set_num_threads(cores)
    m = importlib.import_module(f"dvg_amd.models.{model}_64")
    esd = params.fill_state_dict(m.encoder(90, 1).state_dict(), 1)
    dsd = params.fill_state_dict(m.decoder(90, 1).state_dict(), 2,
                                 params.decoder_transposed_keys(m.decoder(90, 1).state_dict(), model))
    from dvg_amd.models.lstm import lstm
    lsd = params.fill_state_dict(lstm(90, 90, 256, 2, batch).state_dict(), 3)
    gsd, lik = params.gp_state(4)
    seq = SyntheticMovingMNIST(seq_len=n_eval, seed=seed).batch(batch)
    x = orc.normalize_data(seq)
    if model == "vgg":
        enc = lambda t: orc.vgg_encoder(t, esd, False)          # noqa: E731
        dec = lambda v, s: orc.vgg_decoder(v, s, dsd, False)    # noqa: E731
    else:
        enc = lambda t: orc.dcgan_encoder(t, esd, False)        # noqa: E731
        dec = lambda v, s: orc.dcgan_decoder(v, s, dsd, False)  # noqa: E731
    eps = {i: params.normal(50 + i, 90, batch) for i in orc.gp_trigger_steps(n_past, n_eval)}
    with torch.no_grad():
        enc(x[0])  # warm the thread pool / allocator
        n, t0 = 0, time.perf_counter()
        while True:   # bounded sample: whole rollouts until ~10 s of CPU work have been timed
            orc.rollout(x, enc, dec, lsd, gsd, lik, n_past, n_eval, eps)
            n += 1
            dt = time.perf_counter() - t0
            if dt >= 10.0 or n >= 40:
                break
    return {"value": round(n * batch * (n_eval - n_past) / dt, 2), "unit": "frames/s", "cores": cores,
            "kind": "port", "sample": f"{n} rollout(s) of the same workload ({model}_64, B={batch}, "
                                      f"{n_past}-in/{n_eval - n_past}-out) = {dt:.1f} s of CPU work, torch-CPU fp32, "
                                      f"{cores} threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="vgg", choices=["vgg", "dcgan"])
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--n_past", type=int, default=10)
    ap.add_argument("--n_future", type=int, default=10)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            print(f"bench.py: --gpus {args.gpus} needs torch.distributed.run with {args.gpus} ranks", file=sys.stderr)
            sys.exit(2)
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from dvg_amd import fused as fused_mod
    from dvg_amd import ops, utils
    from dvg_amd.data import SyntheticMovingMNIST
    from dvg_amd.rollout import GraphedRollout, sample_rollout

    n_eval = args.n_past + args.n_future
    enc, dec, fp, gp, lik = build_models(args.model, args.batch, 1, dev, args.seed + rank)
    # inputs resident in HBM before the timed region; composited on the GPU (identical to normalize_data(host batch))
    x = SyntheticMovingMNIST(seq_len=n_eval, seed=args.seed + rank).batch_device(args.batch, dev)
    x = [t.to(dev) for t in x]
    calibrate_batchnorm(enc, dec, x[0])

    def eager_step():
        return sample_rollout(enc, dec, fp, gp, lik, x, args.n_past, n_eval)

    if args.no_graph:
        step = eager_step
    else:   # the whole rollout as ONE hipGraph replay
        graphed = GraphedRollout(enc, dec, fp, gp, lik, x, args.n_past, n_eval)
        step = graphed

    for _ in range(args.warmup):
        step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert bool(torch.isfinite(out[-1]).all())

    frames = args.batch * args.n_future * args.steps * world
    result = {
        "metric": "predicted frames/sec at 64x64, batch 64, 10-in/10-out",
        "value": round(frames / dt, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"Moving-MNIST 64x64 rollout (generate_frames.py make_gifs sample loop), "
                               f"{args.model}_64 + lstm + GP trigger sample at i%15==0, batch {args.batch} per GPU, "
                               f"{args.n_past}-in/{args.n_future}-out", "model_family": args.model,
                   "batch_per_gpu": args.batch, "n_past": args.n_past, "n_future": args.n_future,
                   "parallelism": f"replicas x{world} (no data-path collective)",
                   "launch": "eager" if args.no_graph else "hipGraph replay",
                   # the skip tensors are frozen after the conditioning frames: the skip half of each decoder block's
                   # first conv is computed once per rollout and added in the epilogue (DVG_SKIP_HOIST=0: recompute)
                   "loop_invariant_skip_halves": "hoisted" if fused_mod.SKIP_HOIST else "recomputed every step",
                   # nearest-x2 upsample + conv3x3 == 4x4 stride-2 transposed conv with K4 = W (*) ones(2x2): the x half of
                   # the decoder blocks' first convs runs with 4/9 of the MACs (DVG_UPCONV_AS_CONVT=0: 9-tap form)
                   "upsample_conv3x3": "as transposed 4x4/s2 conv" if (fused_mod.SKIP_HOIST and fused_mod.UPCONV_AS_CONVT)
                   else "9-tap conv on the upsampled grid"},
    }

    if rank == 0:
        # roofline leg: the same rollout with every launch bracketed by HIP events on the launch stream
        timer = ops.KernelTimer()
        ops.set_timer(timer)
        for _ in range(3):
            eager_step()   # events need eager launches; same kernels, same shapes as the graphed step
        ops.set_timer(None)
        agg = timer.summary()
        total_ms = sum(a["ms"] for a in agg.values())
        dom = max(agg, key=lambda k: agg[k]["ms"])
        a = agg[dom]
        ach = a["flops"] / (a["ms"] * 1e-3) / 1e12
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside bench.py)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", {"conv3x3_igemm": "r01_conv3x3_traffic.json",
                                                "conv4x4s2_igemm": "r01_conv4x4s2_traffic.json"}.get(dom, "-"))
        if args.batch == 64 and os.path.exists(tpath):
            traffic = json.load(open(tpath))["traffic_bytes_per_launch"]
        result["roofline"] = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                              "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                              "algorithmic_bytes_per_launch": round(a["bytes"] / a["launches"]),
                              "algorithmic_flops_per_launch": round(a["flops"] / a["launches"]),
                              "launches_per_step": a["launches"] // 3,
                              "avg_launch_us": round(1000 * a["ms"] / a["launches"], 2),
                              "share_of_kernel_time": round(a["ms"] / total_ms, 4)}
        result["kernels"] = {k: {"launches_per_step": v["launches"] // 3,
                                 "avg_us": round(1000 * v["ms"] / v["launches"], 2),
                                 "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                                 "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                                 # fraction of the 8 TB/s HBM peak on the kernel's ALGORITHMIC bytes (meaningful for the
                                 # HBM-bound first / last layers; the MFMA-bound kernels sit far below by design)
                                 "hbm_frac": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
                             for k, v in agg.items()}
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(args.model, args.batch, args.n_past, n_eval, args.seed)
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
