#!/usr/bin/env python3
"""bench.py — predicted frames/s of the DVG inference rollout on MI355X.

A "step" is ONE rollout sample (one pass of the make_gifs sample loop, generate_frames.py:143-177)
over one batch of synthetic Moving-MNIST: B=64, 64x64, 10 conditioning + 10 predicted frames
(19 encoder calls, 10 decoder calls, 19 LSTM steps, 1 GP trigger sample at i=15), i.e.
BASELINE.json configs[1].  value = B * n_future * steps * n_gpus / time  (vgg_64; the dcgan_64 family of
the same metric is measured in the same process and reported under "families").

  python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  When the driver launches the ranks itself (torch.distributed.run sets
WORLD_SIZE) each rank runs main(); from a bare shell `python bench.py --gpus N` spawns that launcher as a
CHILD process before anything touches a GPU and relays rank 0's JSON line and the exit code.
The rollout shards by replicas (every GPU rolls out its own batch: no data-path collective), scaling = "weak".
The path that does have an exchange step - data-parallel training with a gradient all-reduce over xGMI
(train.py:200-248 per step) - is timed as a second leg and reported under "train"
(BASELINE.json configs[3] shape: dcgan_64, nc=3, 16 clips per GPU, 2-in/10-out).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from dvg_amd.benchlib import (  # noqa: E402,F401
    Ctx, PEAK_BF16_MFMA_TFLOPS, PEAK_F32_AS_BF16X3_TFLOPS, PEAK_F32_MFMA_TFLOPS, PEAK_HBM_GBS,
    SURVEY_BYTES_PER_ROLLOUT, SURVEY_FLOPS_PER_ROLLOUT, TRAFFIC_FILES, TRAIN_C4, TRAIN_EXTRA, build_models,
    calibrate_batchnorm, gp_trigger_leg, graphed_train_leg, hbm_bound_layers, make_gifs_leg, measure_train,
    profile_file, rollout_traffic, train_leg, usable_cores,
)

F32MFMA_LIB = os.path.join(ROOT, "dvg_amd", "csrc", "libdvg_hip_f32mfma.so")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="vgg", choices=["vgg", "dcgan"], help="family reported as `value`")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--n_past", type=int, default=10)
    ap.add_argument("--n_future", type=int, default=10)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the oracle check of the frames the timed region produced (rank 0; `check` in the JSON line)")
    ap.add_argument("--sustained-s", type=float, default=5.0,
                    help="seconds of back-to-back rollouts (same graphs) for the `sustained` figure beside `value`; 0 = skip")
    ap.add_argument("--allow-variant", action="store_true",
                    help="emit a line although DVG_HIP_LIB points at another build of the library or the loaded build is a "
                         "timing experiment (X3_TERMS != 6, ABLATE != 0, ...): the line then says so in `build`")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--tile-policy", default="auto", choices=["auto", "latency", "energy"],
                    help="dvg_set_tile_policy of the measured kernels; auto = energy with more than one rollout in flight "
                         "(tools/profile_round.sh: --inflight 1 --tile-policy energy = isolated durations of the headline's kernels)")
    ap.add_argument("--inflight", type=int, default=3,
                    help="complete rollouts in flight at once (independent samples of the make_gifs loop, one hipGraph and one "
                         "stream each); 1 = one serial chain of launches")
    ap.add_argument("--no-families", action="store_true", help="skip the other model family")
    ap.add_argument("--no-roofline", action="store_true", help="skip the eager HIP-event leg (timeline profiling runs)")
    ap.add_argument("--no-train-leg", action="store_true", help="skip the data-parallel training leg")
    ap.add_argument("--no-f32mfma-leg", action="store_true",
                    help="skip the comparison run on the native f32-MFMA build of the library (a child process, N = 1 only)")
    ap.add_argument("--train-iters", type=int, default=6)
    ap.add_argument("--no-train-shapes", action="store_true", help="skip the extra single-GPU training shapes (C2 / C4 vgg / C5)")
    ap.add_argument("--no-make-gifs-leg", action="store_true", help="skip the make_gifs (C3) leg")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the C1 (batch 8, 5-in/5-out) leg, the GPtrigger_gen leg and the isolated HBM-bound layers")
    ap.add_argument("--nsample", type=int, default=12, help="sample rollouts per batch of the make_gifs leg")
    ap.add_argument("--train-full-graph", action="store_true",
                    help="with several ranks, also try the iteration as ONE hipGraph with the RCCL all-reduces captured "
                         "inside.  Off by default: on this stack (PyTorch 2.10 / ROCm 7) the c10d watchdog thread may query a "
                         "collective's event while it is still 'recorded in a capturing stream' (hipErrorCapturedEvent), "
                         "which terminates the process - seen in 1 of 5 runs with a one-rank RCCL group")
    ap.add_argument("--train-graph-timeout", type=float, default=240.0,
                    help="seconds the graphed data-parallel leg may take before it is reported as timed out")
    return ap.parse_args(argv)


def self_launch(args) -> int:
    """`python bench.py --gpus N` from a bare shell: start `torch.distributed.run` with N ranks as a child process
    (never exec: this process has not touched a GPU and will not) and relay its output and exit code."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env)
    last_json = None
    for line in proc.stdout:
        if line.lstrip().startswith("{") and '"metric"' in line:
            last_json = line.strip()
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    if last_json is not None:
        print(last_json, flush=True)
    return rc if (rc != 0 or last_json is not None) else 1


CHECK_BAR = 1e-4   # BASELINE.json north_star: "outputs match the reference CPU path within 1e-4 relative on fp32 frames"


def snapshot_case(model, mods, x, chains, n_past, n_eval):
    """What the oracle needs to redo the rollouts the timed graphs just ran, copied to the host: the modules' state_dicts (the
    SAME weights, BatchNorm running statistics and GP state the GPU path used), the input clips, and per chain the GP base
    samples eps its last replay drew and the frames that replay produced."""
    import torch
    torch.cuda.synchronize()
    cpu = lambda sd: {k: v.detach().cpu().clone() for k, v in sd.items()}   # noqa: E731
    enc, dec, fp, gp, lik = mods
    return {"model": model, "n_past": n_past, "n_eval": n_eval, "batch": int(x[0].shape[0]),
            "sds": (cpu(enc.state_dict()), cpu(dec.state_dict()), cpu(fp.state_dict()), cpu(gp.state_dict()), cpu(lik.state_dict())),
            "x": [t.detach().cpu().clone() for t in x],
            "chains": [{"eps": {i: e.detach().cpu().clone() for i, e in eps.items()},
                        "frames": [f.detach().cpu().clone() for f in frames]} for eps, frames in chains]}


def check_against_oracle(case, oracle_frames):
    """max |a - b| / max |b| per frame (tests/common.rel_err) of the frames the TIMED graphs produced against the oracle's
    rollout of the same weights, clips and GP base samples, over every predicted frame of every chain the oracle re-ran."""
    n_past, n_eval = case["n_past"], case["n_eval"]
    per_chain = []
    for k, ref in sorted(oracle_frames.items()):
        got = case["chains"][k]["frames"]
        errs = []
        for t in range(n_past, n_eval):
            a, b = got[t].double(), ref[t].double()
            errs.append(float((a - b).abs().max() / max(float(b.abs().max()), 1e-12)))
        per_chain.append(max(errs))
    worst = max(per_chain) if per_chain else None
    return {"rel_err_vs_oracle": None if worst is None else float(f"{worst:.3e}"), "bar": CHECK_BAR,
            "ok": bool(worst is not None and worst < CHECK_BAR),
            "frames_compared": len(per_chain) * (n_eval - n_past) * case["batch"],
            "chains_compared": len(per_chain), "per_chain": [float(f"{e:.3e}") for e in per_chain],
            "what": "the predicted frames left in the static buffers by the LAST replay of each timed hipGraph chain against "
                    "oracle.rollout (generate_frames.py:143-177 restated on torch-CPU fp32, GP in fp64) with the same weights, "
                    "clips and GP base samples; rel_err = max|a-b| / max|b| per frame, worst frame reported"}


def cpu_baseline(model: str, batch: int, n_past: int, n_eval: int, seed: int, budget_s: float = 10.0, case=None,
                 max_chains=None):
    """The oracle (CPU restatement, parity-checked against the reference's modules) timed on the host
    cores of this box: whole rollouts of the same workload until `budget_s` of CPU work have been timed.
    With `case` (snapshot_case) the rollouts are the ones the timed GPU graphs ran - same weights, clips and GP base samples,
    one chain after the other - and their frames are returned as the checker's reference (second value)."""
    import importlib
    import torch
    from oracle import dvg_oracle as orc
    from oracle import params
    from dvg_amd.data import SyntheticMovingMNIST
    cores = usable_cores()
    torch.set_num_threads(cores)
    if case is not None:
        esd, dsd, lsd, gsd, lik = case["sds"]
        x = case["x"]
        eps_list = [c["eps"] for c in case["chains"]][:max_chains]
    else:
        m = importlib.import_module(f"dvg_amd.models.{model}_64")
        esd = params.fill_state_dict(m.encoder(90, 1).state_dict(), 1)
        dsd = params.fill_state_dict(m.decoder(90, 1).state_dict(), 2,
                                     params.decoder_transposed_keys(m.decoder(90, 1).state_dict(), model))
        from dvg_amd.models.lstm import lstm
        lsd = params.fill_state_dict(lstm(90, 90, 256, 2, batch).state_dict(), 3)
        gsd, lik = params.gp_state(4)
        seq = SyntheticMovingMNIST(seq_len=n_eval, seed=seed).batch(batch)
        x = orc.normalize_data(seq)
        eps_list = [{i: params.normal(50 + i, 90, batch) for i in orc.gp_trigger_steps(n_past, n_eval)}]
    if model == "vgg":
        enc = lambda t: orc.vgg_encoder(t, esd, False)          # noqa: E731
        dec = lambda v, s: orc.vgg_decoder(v, s, dsd, False)    # noqa: E731
    else:
        enc = lambda t: orc.dcgan_encoder(t, esd, False)        # noqa: E731
        dec = lambda v, s: orc.dcgan_decoder(v, s, dsd, False)  # noqa: E731
    frames = {}
    with torch.no_grad():
        enc(x[0])  # warm the thread pool / allocator
        n, t0 = 0, time.perf_counter()
        while True:   # bounded sample: whole rollouts until ~budget_s of CPU work have been timed (and every chain once)
            k = n % len(eps_list)
            out = orc.rollout(x, enc, dec, lsd, gsd, lik, n_past, n_eval, eps_list[k])
            frames.setdefault(k, out)
            n += 1
            dt = time.perf_counter() - t0
            if (dt >= budget_s and n >= len(eps_list)) or n >= 40:
                break
    res = {"value": round(n * batch * (n_eval - n_past) / dt, 2), "unit": "frames/s", "cores": cores,
           "kind": "port", "sample": f"{n} rollout(s) of the same workload ({model}_64, B={batch}, "
                                     f"{n_past}-in/{n_eval - n_past}-out) = {dt:.1f} s of CPU work, torch-CPU fp32, "
                                     f"{cores} threads" + ("; the weights, clips and GP draws of the timed GPU chains "
                                                           "(their frames are the `check` reference)" if case is not None else "")}
    return (res, frames) if case is not None else res


def measure_rollout(ctx: Ctx, args, model: str, steps: int, warmup: int) -> dict:
    """W untimed + K timed rollouts of one family; barrier + synchronize on both sides, max over ranks.
    Rank 0 adds the per-kernel HIP-event leg (roofline of the dominant kernel)."""
    import torch
    from dvg_amd import ops
    from dvg_amd.data import SyntheticMovingMNIST
    from dvg_amd.rollout import ConcurrentRollouts, sample_rollout
    n_eval = args.n_past + args.n_future
    enc, dec, fp, gp, lik = build_models(model, args.batch, 1, ctx.dev, args.seed + ctx.rank)
    # inputs resident in HBM before the timed region; composited on the GPU (identical to normalize_data(host batch))
    x = SyntheticMovingMNIST(seq_len=n_eval, seed=args.seed + ctx.rank).batch_device(args.batch, ctx.dev)
    calibrate_batchnorm(enc, dec, x[0])

    # eager launches (--no-graph, the HIP-event leg): the GP base samples come from a static buffer refilled per step, so that
    # the oracle check can redo the last step's draw
    eager_eps = {i: torch.randn(90, args.batch, device=ctx.dev) for i in range(args.n_past, n_eval) if i % 15 == 0}

    # tile policy of the measured kernels (dvg_set_tile_policy): energy-lean tiles when several rollouts are in flight - also for
    # the eager forms of the same step (--no-graph, the HIP-event leg, the PMC passes of tools/profile_round.sh), so that every
    # figure of a line describes the same kernels
    energy = max(1, args.inflight) > 1 if args.tile_policy == "auto" else args.tile_policy == "energy"

    def eager_step():
        for e in eager_eps.values():
            e.normal_()
        with ops.tile_policy(energy):
            return sample_rollout(enc, dec, fp, gp, lik, x, args.n_past, n_eval, eps_by_step=eager_eps)

    # A step = one COMPLETE rollout (conditioning + prediction, B clips).  The rollouts of the make_gifs sample loop are
    # independent: `inflight` of them run at once, each as its own hipGraph on its own stream (rollout.ConcurrentRollouts);
    # K steps = K rollouts issued round-robin, per-rollout work and results unchanged.
    inflight = 1 if args.no_graph else max(1, args.inflight)
    cr = None if args.no_graph else ConcurrentRollouts(enc, dec, fp, gp, lik, x, args.n_past, n_eval, inflight=inflight,
                                                       energy_tiles=energy)

    def run(n, chains=None):
        if cr is None:
            for _ in range(n):
                last = eager_step()
            return last
        return cr.run(n, chains=chains)[0]

    run(warmup)
    ctx.barrier()
    t0 = time.perf_counter()
    out = run(steps)
    ctx.barrier()
    per_rank = ctx.all_ranks(time.perf_counter() - t0)
    dt = max(per_rank)
    assert bool(torch.isfinite(out[-1]).all())
    frames = args.batch * args.n_future * steps * ctx.world
    res = {"value": round(frames / dt, 1), "ms_per_step": round(1000 * dt / steps, 3), "rollouts_in_flight": inflight,
           # wall time of the K timed steps on every rank, rank order (the reported time is the max)
           "per_rank_ms_per_step": [round(1000 * t / steps, 3) for t in per_rank]}
    if ctx.rank == 0 and not args.no_check:
        # the frames the TIMED region left behind (every chain's last replay) and what the oracle needs to redo them; the
        # comparison itself runs later, on the host, outside every timed region (main: cpu_baseline / check)
        chains = [(r.eps, r.frames) for r in cr.rollouts] if cr is not None else [(eager_eps, out)]
        res["_case"] = snapshot_case(model, (enc, dec, fp, gp, lik), x, chains, args.n_past, n_eval)
    if args.sustained_s > 0 and steps >= 2:
        # the same graphs for >= sustained_s seconds: the chip is power / clock-managed and the K timed steps of the contract
        # are a fraction of a second ("nobody knows whether the figure holds for 10 s")
        n_sus = max(steps, int(args.sustained_s / (dt / steps)) + 1)
        if ctx.dist is not None:     # every rank must run the same count (the barrier pairs bracket it)
            n_sus = int(max(ctx.all_ranks(float(n_sus))))
        ctx.barrier()
        t0 = time.perf_counter()
        run(n_sus)
        ctx.barrier()
        dts = ctx.max_over_ranks(time.perf_counter() - t0)
        res["sustained"] = {"value": round(args.batch * args.n_future * n_sus * ctx.world / dts, 1), "steps": n_sus,
                            "seconds": round(dts, 2), "ms_per_step": round(1000 * dts / n_sus, 3),
                            "ratio_to_value": round((args.batch * args.n_future * n_sus * ctx.world / dts) / (frames / dt), 4)}
    if inflight > 1:
        # the same K rollouts as ONE serial chain of launches (one graph, one stream), for comparison: a graph of its own,
        # captured under the LATENCY tile policy (what a caller with one chain gets: rollout.GraphedRollout's default; the
        # chains above carry the energy-lean tiles, which make one chain 3 % slower and three in flight 4.6 % faster)
        from dvg_amd.rollout import GraphedRollout
        one = GraphedRollout(enc, dec, fp, gp, lik, x, args.n_past, n_eval)
        one()
        ctx.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            one()
        ctx.barrier()
        dt1 = ctx.max_over_ranks(time.perf_counter() - t0)
        res["single_chain"] = {"value": round(frames / dt1, 1), "ms_per_step": round(1000 * dt1 / steps, 3),
                               "tile_policy": "latency"}
        ctx.barrier()
        t0 = time.perf_counter()
        run(steps, chains=1)     # ... and one of the in-flight chains' graphs (energy-lean tiles) alone
        ctx.barrier()
        dt2 = ctx.max_over_ranks(time.perf_counter() - t0)
        res["single_chain"]["energy_tiles_ms_per_step"] = round(1000 * dt2 / steps, 3)
        del one
    if ctx.rank != 0 or args.no_roofline:
        return res
    # roofline leg: the same rollout with every launch bracketed by HIP events on the launch stream
    timer = ops.KernelTimer()
    ops.set_timer(timer)
    for _ in range(3):
        eager_step()   # events need eager launches; same kernels (tile policy included), same shapes as the graphed step
    ops.set_timer(None)
    agg = timer.summary()
    total_ms = sum(a["ms"] for a in agg.values())
    dom = max(agg, key=lambda k: agg[k]["ms"])
    a = agg[dom]
    ach = a["flops"] / (a["ms"] * 1e-3) / 1e12
    traffic, tsrc = None, None
    tname = TRAFFIC_FILES.get((model, dom))
    tfile = profile_file(tname) if tname else None
    if tfile and args.batch == 64:
        traffic = json.load(open(tfile))["traffic_bytes_per_launch"]
        tsrc = f"profiles/{os.path.basename(tfile)}: committed rocprofv3 PMC passes of this command (constant, NOT measured by this run)"
    from dvg_amd import _lib
    x3 = _lib.lib().dvg_mfma_mode() == 1 and dom in ("winograd_gemm", "conv3x3_igemm", "conv4x4s2_igemm", "convT4x4s2_igemm")
    mfma_peak = PEAK_F32_AS_BF16X3_TFLOPS if x3 else PEAK_F32_MFMA_TFLOPS
    # WHICH roof binds follows from the kernel's own executed flops and algorithmic bytes: its HBM roof in TFLOP/s is
    # intensity x 8 TB/s, and the lower of the two roofs is the bound (the Winograd GEMM's 45 FLOP/B put its
    # HBM roof, 364 TF, below the 419 TF of the bf16-triple matrix pipe).  `achieved` / `peak` are in the binding roof's unit.
    sec = a["ms"] * 1e-3
    gbs = a["bytes"] / sec / 1e9
    hbm_roof_tflops = a["flops"] / a["bytes"] * PEAK_HBM_GBS * 1e9 / 1e12
    bound = "hbm" if hbm_roof_tflops < mfma_peak else "mfma"
    if bound == "hbm":
        r_ach, r_peak, r_unit = gbs, PEAK_HBM_GBS, "GB/s"
    else:
        r_ach, r_peak, r_unit = ach, mfma_peak, "TFLOP/s"
    res["roofline"] = {"kernel": dom, "bound": bound, "achieved": round(r_ach, 2), "peak": round(r_peak, 1),
                       "unit": r_unit, "frac": round(r_ach / r_peak, 4), "traffic": traffic,
                       "traffic_source": tsrc,
                       "bound_is": (f"the lower of the kernel's two roofs: intensity {a['flops'] / a['bytes']:.1f} FLOP/B x 8 TB/s = "
                                    f"{hbm_roof_tflops:.0f} TFLOP/s (HBM) against {mfma_peak:.0f} TFLOP/s (matrix pipe)"),
                       "mfma": {"achieved_tflops": round(ach, 2), "peak_tflops": round(mfma_peak, 1), "frac": round(ach / mfma_peak, 4)},
                       "hbm": {"achieved_gbs": round(gbs, 1), "peak_gbs": PEAK_HBM_GBS, "frac": round(gbs / PEAK_HBM_GBS, 4),
                               "bytes_are": "ALGORITHMIC (operands + result of a launch, each once)"},
                       # the committed PMC traffic per launch over THIS run's event time per launch, against the 8 TB/s peak
                       "hbm_frac_counter": None if traffic is None else
                       round(traffic / (sec / a["launches"]) / 1e9 / PEAK_HBM_GBS, 4),
                       "achieved_is": "EXECUTED fp32 flops of the kernel (2 x its multiply-adds, each counted once) / its HIP-event time"
                                      if bound == "mfma" else "ALGORITHMIC bytes of the kernel's launches / their HIP-event time",
                       "peak_is": ("HBM3E peak (MI355X_MICROARCH.md: 8 TB/s spec, 6.3 achievable)" if bound == "hbm" else
                                   "bf16 dense MFMA peak / 6: the kernel forms each fp32 product slab as six v_mfma_f32_32x32x16_bf16 "
                                   "on exact bf16 triples, so `frac` is also the busy fraction of the bf16 matrix pipe"
                                   if x3 else "f32-input dense MFMA peak (v_mfma_f32_32x32x2_f32)"),
                       # the same achieved rate against the NATIVE f32-MFMA roof (the peak r01 / r02 lines were priced on): above
                       # 1 means the kernel beats anything the f32-input MFMA could do
                       "frac_of_f32_mfma_peak": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                       "algorithmic_bytes_per_launch": round(a["bytes"] / a["launches"]),
                       "executed_flops_per_launch": round(a["flops"] / a["launches"]),
                       "algorithmic_flops_per_launch": round(a["alg_flops"] / a["launches"]),
                       "launches_per_step": a["launches"] // 3,
                       "avg_launch_us": round(1000 * a["ms"] / a["launches"], 2),
                       "share_of_kernel_time": round(a["ms"] / total_ms, 4),
                       # SURVEY.md 8(d)'s DIRECT-FORM count of the same launches / the same time: may exceed 1 (Winograd
                       # executes 1/4 of the direct form's multiplies) - labelled, never the headline `frac`
                       "frac_direct_form": round(a["alg_flops"] / (a["ms"] * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                       # sum of the HIP-event durations of ALL launches of one step (one chain, eager): against
                       # `single_chain.ms_per_step` it shows the launch gaps, against `ms_per_step` what the chains overlap
                       "kernel_time_sum_ms": round(total_ms / 3, 3),
                       # share of that sum spent in the HBM-bound Winograd transform passes (input / output / fused)
                       "transform_share": round(sum(v["ms"] for k, v in agg.items()
                                                    if k.startswith("winograd_") and k != "winograd_gemm") / total_ms, 4)}
    # The 3x3 layers as a whole (SURVEY 8(d) counts a layer's direct-form FLOPs): direct implicit-GEMM launches plus, for the
    # layers that run as Winograd F(4x4) / F(2x2), their transform + batched-GEMM launches.  Can exceed the fp32 MFMA peak:
    # Winograd executes 1/4 (1/2.25) of the direct form's multiplies.
    fam = [k for k in agg if k == "conv3x3_igemm" or k.startswith("winograd_")]     # (r04 left the pool hand-over out)
    if any(k.startswith("winograd") for k in fam):
        ms = sum(agg[k]["ms"] for k in fam)
        alg = sum(agg[k]["alg_flops"] for k in ("conv3x3_igemm", "winograd_gemm") if k in agg)
        exe = sum(agg[k]["flops"] for k in ("conv3x3_igemm", "winograd_gemm") if k in agg)
        res["roofline"]["conv3x3_layers"] = {
            "launches_per_step": sum(agg[k]["launches"] for k in fam) // 3, "ms_per_step": round(ms / 3, 3),
            "algorithmic_tflops": round(alg / (ms * 1e-3) / 1e12, 2), "executed_tflops": round(exe / (ms * 1e-3) / 1e12, 2),
            "algorithmic_frac_of_fp32_mfma_peak": round(alg / (ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
            "share_of_kernel_time": round(ms / total_ms, 4)}
    # the rollout as a whole against the fp32 MFMA roof: FLOPs its launches EXECUTE per step, and the step's direct-form
    # ("algorithmic", SURVEY.md 8(d): 2 x MACs of every conv / convT / linear as the reference runs them, skip halves
    # recomputed every step, 9-tap upsample convs) FLOPs, both over the timed wall time per step
    exe_step = sum(v["flops"] for v in agg.values()) / 3
    alg_step = SURVEY_FLOPS_PER_ROLLOUT.get(model) if (args.batch, args.n_past, args.n_future) == (64, 10, 10) else None
    sec = res["ms_per_step"] * 1e-3
    res["rollout_flops"] = {"executed_per_step": round(exe_step), "executed_tflops": round(exe_step / sec / 1e12, 2),
                            "executed_frac_of_fp32_mfma_peak": round(exe_step / sec / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                            "algorithmic_per_step": alg_step,
                            "algorithmic_tflops": None if alg_step is None else round(alg_step / sec / 1e12, 2),
                            "algorithmic_frac_of_fp32_mfma_peak": None if alg_step is None else
                            round(alg_step / sec / 1e12 / PEAK_F32_MFMA_TFLOPS, 4)}
    rt = rollout_traffic(model, args, res["ms_per_step"])
    if rt is not None:
        res["rollout_traffic"] = rt
    res["kernels"] = {k: {"launches_per_step": v["launches"] // 3,
                          "avg_us": round(1000 * v["ms"] / v["launches"], 2),
                          "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2),
                          "gbs": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1),
                          # fraction of the 8 TB/s HBM peak on the kernel's ALGORITHMIC bytes (meaningful for the
                          # HBM-bound first / last layers; the MFMA-bound kernels sit far below by design)
                          "hbm_frac": round(v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}
                      for k, v in agg.items()}
    return res


def c1_leg(ctx: Ctx, args) -> dict:
    """BASELINE.json configs[0] (C1): Moving-MNIST 64x64, batch 8, 5-in/5-out, vgg_64 + lstm - the configuration BASELINE.md names
    as the reference's CPU-runnable case.  The same rollout measurement as the headline at that shape (hipGraph replay, three
    rollouts in flight and one serial chain) and, on rank 0 at N = 1, the CPU oracle at the SAME configuration on this box."""
    import copy
    a = copy.copy(args)
    a.batch, a.n_past, a.n_future = 8, 5, 5
    a.no_roofline = True
    a.sustained_s = 0.0
    r = measure_rollout(ctx, a, "vgg", max(20, args.steps), args.warmup)
    out = {"workload": "Moving-MNIST 64x64 rollout, vgg_64 + lstm + GP (no trigger step inside 5 + 5), batch 8 per GPU, 5-in/5-out "
                       "(BASELINE.json configs[0])",
           "value": r["value"], "unit": "frames/s", "ms_per_step": r["ms_per_step"], "single_chain": r.get("single_chain"),
           "rollouts_in_flight": r["rollouts_in_flight"]}
    case = r.pop("_case", None)
    if ctx.rank == 0 and ctx.world == 1 and not args.no_cpu_baseline:
        if case is not None:
            out["cpu_baseline"], ref = cpu_baseline("vgg", 8, 5, 10, args.seed, budget_s=4.0, case=case)
            out["check"] = check_against_oracle(case, ref)
        else:
            out["cpu_baseline"] = cpu_baseline("vgg", 8, 5, 10, args.seed, budget_s=4.0)
        out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    return out


def f32mfma_leg(args) -> dict:
    """The SAME rollout measurement on the native f32-MFMA build of the library (libdvg_hip_f32mfma.so, `make f32mfma`),
    in a child process (one process holds one build of the library), after this process's own timed region: what the bf16-triple
    arithmetic buys, measured on the same box minutes apart.  Reported beside `value`, never as `value`."""
    import subprocess
    if not os.path.exists(F32MFMA_LIB):
        return {"skipped": "dvg_amd/csrc/libdvg_hip_f32mfma.so not built (make -C dvg_amd/csrc f32mfma)"}
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
           "--model", args.model, "--batch", str(args.batch), "--n_past", str(args.n_past), "--n_future", str(args.n_future),
           "--seed", str(args.seed), "--inflight", str(args.inflight), "--no-families", "--no-cpu-baseline", "--no-train-leg",
           "--no-f32mfma-leg", "--no-make-gifs-leg", "--no-extra-legs", "--sustained-s", "0"] + (["--no-graph"] if args.no_graph else []) + \
        (["--no-check"] if args.no_check else [])
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["DVG_HIP_LIB"] = F32MFMA_LIB
    try:
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    except subprocess.TimeoutExpired:
        return {"skipped": "timed out"}
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        return {"skipped": f"child exited {r.returncode}: {r.stderr[-300:]}"}
    d = json.loads(lines[-1])
    rf = d.get("roofline", {})
    return {"library": "dvg_amd/csrc/libdvg_hip_f32mfma.so (-DDVG_BF16X3=0)", "value": d["value"], "ms_per_step": d["ms_per_step"],
            "single_chain": d.get("single_chain"), "build": (d.get("build") or {}).get("raw"), "check": d.get("check"),
            "roofline": {k: rf.get(k) for k in ("kernel", "achieved", "peak", "frac", "avg_launch_us", "launches_per_step",
                                                "kernel_time_sum_ms", "transform_share")}}


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    import torch
    world_env = int(os.environ.get("WORLD_SIZE", "1"))
    if world_env != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world_env}", file=sys.stderr)
        sys.exit(2)
    if os.environ.get("DVG_DP_SHARE_GPU") != "1" and torch.cuda.device_count() < int(os.environ.get("LOCAL_RANK", "0")) + 1:
        print(f"bench.py: rank needs GPU {os.environ.get('LOCAL_RANK', '0')}, only {torch.cuda.device_count()} visible",
              file=sys.stderr)
        sys.exit(2)
    ctx = Ctx(args)

    from dvg_amd import _lib, fused as fused_mod
    # WHICH build of the library is loaded, before anything is measured: a headline only from the product build (the default
    # path - or, in the f32-MFMA comparison child, the in-tree f32 build - with no timing-experiment knob)
    build = _lib.build_info()
    if not build["product"] and not args.allow_variant:
        print(f"bench.py: refusing to measure {build['path']} ({build['raw']}): not the product build of libdvg_hip.so "
              "(DVG_HIP_LIB set to a variant, or a timing-experiment build).  --allow-variant to run anyway.", file=sys.stderr)
        sys.exit(4)
    main_res = measure_rollout(ctx, args, args.model, args.steps, args.warmup)
    main_case = main_res.pop("_case", None)
    result = {
        "metric": "predicted frames/sec at 64x64, batch 64, 10-in/10-out",
        "value": main_res["value"], "unit": "frames/s", "n_gpus": ctx.world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": main_res["ms_per_step"], "higher_is_better": True,
        "single_chain": main_res.get("single_chain"), "per_rank_ms_per_step": main_res.get("per_rank_ms_per_step"),
        "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        # the same graphs for >= 5 s (the chip is power-managed; the K timed steps are a fraction of a second)
        "sustained": main_res.get("sustained"),
        # dvg_build_info() of the loaded library: arithmetic, timing-experiment knobs (must be 6 / 0 / 0 / 0), source hash
        "build": {k: build[k] for k in ("raw", "product", "from_env")} | {"path": os.path.relpath(build["path"], ROOT)},
        # how the implicit-GEMM kernels form their fp32 products (dvg_mfma_mode(), include/dvg_hip.h): inputs, outputs, weights,
        # accumulators and every other kernel are fp32 in both builds
        "arithmetic": ("fp32 operands split exactly into three bf16 terms, six bf16 MFMAs per K=16 slab, fp32 accumulate "
                       "(round-to-nearest split, dropped cross terms < 2^-24 |a||b|; measured error vs fp64 <= the f32 MFMA's)"
                       if _lib.lib().dvg_mfma_mode() == 1 else "native f32-input MFMA"),
        "config": {"workload": f"Moving-MNIST 64x64 rollout (generate_frames.py make_gifs sample loop), "
                               f"{args.model}_64 + lstm + GP trigger sample at i%15==0, batch {args.batch} per GPU, "
                               f"{args.n_past}-in/{args.n_future}-out", "model_family": args.model,
                   "batch_per_gpu": args.batch, "n_past": args.n_past, "n_future": args.n_future,
                   "parallelism": f"replicas x{ctx.world} (no data-path collective)",
                   "launch": "eager" if args.no_graph else "hipGraph replay",
                   # independent rollouts (samples of the make_gifs loop) in flight at once, one hipGraph + stream each; every
                   # step is a complete rollout, ms_per_step = wall time / steps; `single_chain` = the same steps back to back
                   "rollouts_in_flight": main_res["rollouts_in_flight"],
                   # dvg_set_tile_policy: several chains in flight keep the board at its power cap, their graphs carry the
                   # energy-lean tiles (same results to fp32 rounding); `single_chain` is a graph with the latency tiles
                   "tile_policy": (("energy" if max(1, args.inflight) > 1 else "latency") if args.tile_policy == "auto"
                                   else args.tile_policy),
                   "step": "one COMPLETE rollout (conditioning + prediction) of one batch; K steps = K rollouts, independent of "
                           "each other (samples of make_gifs' nsample loop), issued round-robin over the chains",
                   # the skip tensors are frozen after the conditioning frames: the skip half of each decoder block's
                   # first conv is computed once per rollout and added in the epilogue (DVG_SKIP_HOIST=0: recompute)
                   "loop_invariant_skip_halves": "hoisted" if fused_mod.SKIP_HOIST else "recomputed every step",
                   # nearest-x2 upsample + conv3x3 == 4x4 stride-2 transposed conv with K4 = W (*) ones(2x2): the x half of
                   # the decoder blocks' first convs runs with 4/9 of the MACs (DVG_UPCONV_AS_CONVT=0: 9-tap form)
                   "upsample_conv3x3": "as transposed 4x4/s2 conv" if (fused_mod.SKIP_HOIST and fused_mod.UPCONV_AS_CONVT)
                   else "9-tap conv on the upsampled grid",
                   # eval-mode 3x3 layers on maps up to 32x32 with >= 64 input channels run as Winograd F(4x4,3x3) (F(2x2) where only
                   # its tile count fits): fp32 data and transforms, 4x / 2.25x fewer multiplies (DVG_WINOGRAD=0: direct form everywhere)
                   "conv3x3_deep_layers": {0: "direct implicit GEMM", 2: "Winograd F(2x2,3x3)", 4: "Winograd F(4x4,3x3) / F(2x2,3x3)"}[
                       fused_mod.WINOGRAD]},
    }
    if ctx.rehearsal:
        result["rehearsal"] = ("DVG_DP_SHARE_GPU=1: all ranks share GPU 0 (backend %s) - a rehearsal of the multi-rank "
                               "control flow, NOT a measurement" % getattr(ctx, "backend", "none"))
    for k in ("roofline", "rollout_flops", "rollout_traffic", "kernels"):
        if k in main_res:
            result[k] = main_res[k]

    def emit():
        if ctx.rank == 0:
            print(json.dumps(result), flush=True)

    if not args.no_families:
        other = "dcgan" if args.model == "vgg" else "vgg"
        fam = measure_rollout(ctx, args, other, args.steps, args.warmup)
        fam_case = fam.pop("_case", None)
        fam["workload"] = result["config"]["workload"].replace(f"{args.model}_64", f"{other}_64")
        result["families"] = {other: fam}
    if not args.no_make_gifs_leg and not args.no_graph:
        ctx.barrier()
        result["make_gifs"] = {m: make_gifs_leg(ctx, args, m, args.nsample)
                               for m in ([args.model] if args.no_families else ["vgg", "dcgan"])}
    if not args.no_extra_legs and not args.no_graph:
        ctx.barrier()
        result["c1"] = c1_leg(ctx, args)
        result["gp_trigger_gen"] = {m: gp_trigger_leg(ctx, args, m) for m in ([args.model] if args.no_families else ["vgg", "dcgan"])}
        if ctx.rank == 0 and "roofline" in result:
            result["roofline"]["hbm_bound_layers"] = hbm_bound_layers(ctx, args)
    n_eval = args.n_past + args.n_future
    # cpu_baseline + check: the oracle re-runs, on the host and outside every timed region, the very rollouts the timed graphs
    # ran (same weights, clips, GP draws).  Its wall time is the CPU baseline (rank 0, N = 1), its frames the checker of the
    # frames the timed region left in the graphs' static buffers.  N > 1: one chain of rank 0, untimed.
    checks_ok = True
    if ctx.rank == 0:
        timed_cpu = ctx.world == 1 and not args.no_cpu_baseline
        jobs = [(args.model, main_case, result, 10.0)]
        if "families" in result:
            jobs += [(other, fam_case, fam, 4.0) for other, fam in result["families"].items()]
        for model, case, sink, budget in jobs:
            if case is None and timed_cpu:
                sink["cpu_baseline"] = cpu_baseline(model, args.batch, args.n_past, n_eval, args.seed, budget_s=budget)
            elif case is not None:
                base, ref = cpu_baseline(model, args.batch, args.n_past, n_eval, args.seed, budget_s=budget if timed_cpu else 0.0,
                                         case=case, max_chains=None if timed_cpu else 1)
                if timed_cpu:
                    sink["cpu_baseline"] = base
                sink["check"] = check_against_oracle(case, ref)
                checks_ok = checks_ok and sink["check"]["ok"]
    if "c1" in result and "check" in result["c1"]:
        checks_ok = checks_ok and result["c1"]["check"]["ok"]
    # under rocprofv3 the child would inherit the preloaded tool: its f32-MFMA kernels would land in the same output
    # directory and pollute the per-kernel statistics / PMC sums of THIS command - skipped there
    profiled = any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_REGISTER_FORCE_LOAD")) or \
        "rocprof" in os.environ.get("LD_PRELOAD", "")
    if (ctx.rank == 0 and ctx.world == 1 and not args.no_f32mfma_leg and _lib.lib().dvg_mfma_mode() == 1
            and "DVG_HIP_LIB" not in os.environ):
        result["f32_mfma_build"] = {"skipped": "running under a profiler"} if profiled else f32mfma_leg(args)
    if not args.no_train_leg:
        ctx.barrier()
        result["train"] = train_leg(ctx, args)
        graphed_train_leg(ctx, args, result, emit)
    emit()
    if ctx.dist is not None:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()
    if not checks_ok:      # the line is out (with check.ok = false in it); a wrong-frames run is not a measurement
        print("bench.py: the timed frames differ from the oracle's by more than 1e-4 (see `check` in the JSON line)", file=sys.stderr)
        sys.exit(5)


if __name__ == "__main__":
    main()
