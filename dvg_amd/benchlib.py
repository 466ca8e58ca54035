"""What bench.py (repo root) is made of besides its headline measurement: the peaks and SURVEY constants the roofline is priced
against, rank / device / process-group context, model construction, and the secondary legs of the JSON line - make_gifs (C3),
the isolated HBM-bound layers, GPtrigger_gen, and data-parallel training (C4 / C5 shapes).  bench.py keeps the argument
surface, the timed rollout, the oracle check and the CPU baseline (the only importers of oracle/ outside tests/)."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E ~8 TB/s
PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2516.6  # MI355X_MICROARCH.md: ~2.5 PF dense = 16 x the f32 MFMA rate (v_mfma_f32_32x32x16_bf16)
# the default library forms every fp32 product slab as SIX bf16 MFMAs on exact bf16 triples (include/dvg_hip.h, ABI 7): the
# roof of an fp32 kernel in that formulation is the bf16 peak / 6
PEAK_F32_AS_BF16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0
# HBM bytes per launch of the dominant kernel come from rocprofv3 PMC passes of this same command, committed under
# profiles/ (rocprofv3 cannot run inside bench.py); the JSON line says so in `traffic_source`.
# direct-form FLOPs of one B = 64, 10-in/10-out rollout (19 encoder + 10 decoder passes + 19 LSTM steps)
SURVEY_FLOPS_PER_ROLLOUT = {"vgg": 4.99e12, "dcgan": 0.51e12}
TRAFFIC_FILES = {("vgg", "conv3x3_igemm"): "conv3x3_traffic.json",
                 ("vgg", "winograd_gemm"): "winograd_gemm_traffic.json",
                 ("dcgan", "conv4x4s2_igemm"): "conv4x4s2_traffic.json",
                 ("dcgan", "convT4x4s2_igemm"): "convT4x4s2_traffic.json"}
# algorithmic bytes of one B = 64, 10-in/10-out rollout (every elementwise op fused)
SURVEY_BYTES_PER_ROLLOUT = {"vgg": 15.7e9, "dcgan": 2.68e9}


def profile_file(name: str):
    """The newest committed profiles/rNN_<name> (rocprofv3 PMC passes cannot run inside bench.py: committed constants)."""
    import glob
    hits = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + name)))
    return hits[-1] if hits else None


def rollout_traffic(model: str, args, ms_per_step: float):
    """HBM-side bytes of ONE rollout from the committed per-kernel PMC sums (profiles/rNN_pmc_by_kernel.json: FETCH_SIZE
    doubled per the gfx950 note + WRITE_SIZE, summed over every kernel of the rollouts profiled) against SURVEY 8(d)'s
    algorithmic bytes; `counter_gbs` prices the counter bytes on THIS run's time per step."""
    f = profile_file("pmc_by_kernel.json")
    alg = SURVEY_BYTES_PER_ROLLOUT.get(model) if (args.batch, args.n_past, args.n_future) == (64, 10, 10) else None
    if f is None or alg is None:
        return None
    d = (json.load(open(f)).get(model) or {}).get("_rollout")
    if not d:
        return None
    cb = d["traffic_bytes_per_rollout"]
    return {"counter_bytes_per_step": round(cb), "algorithmic_bytes_per_step": round(alg), "ratio": round(cb / alg, 3),
            "counter_gbs": round(cb / (ms_per_step * 1e-3) / 1e9, 1),
            "counter_frac_of_hbm_peak": round(cb / (ms_per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4),
            "source": f"profiles/{os.path.basename(f)}: committed rocprofv3 PMC passes of this command (constant, NOT measured "
                      "by this run; Infinity-Cache hits are counted, MI355X_MICROARCH.md)"}


# ------------------------------------------------------------------------------------------------------------
def build_models(model: str, batch: int, nc: int, dev, seed: int):
    import importlib
    import torch
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


def calibrate_batchnorm(enc, dec, frame):
    """Give the BatchNorm layers the running statistics a trained model would have (batch statistics
    of the synthetic data, momentum 1) so that eval-mode activations stay O(1) through all layers
    instead of collapsing / exploding with the N(0,0.02) init (degenerate operands flatter DVFS)."""
    import torch
    with torch.no_grad():
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


class Ctx:
    """rank / world / device / torch.distributed handle of this process."""

    def __init__(self, args):
        import torch
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local = int(os.environ.get("LOCAL_RANK", "0"))
        # Rehearsal of the multi-rank control flow on a ONE-GPU box (never a measurement): DVG_DP_SHARE_GPU=1 puts every
        # rank on device 0 and DVG_DP_BACKEND=gloo replaces RCCL (which refuses two ranks on one device).
        self.rehearsal = os.environ.get("DVG_DP_SHARE_GPU") == "1"
        if self.rehearsal:
            self.local = 0
        assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
        torch.cuda.set_device(self.local)
        self.dev = torch.device("cuda", self.local)
        self.dist = None
        if self.world > 1 or os.environ.get("DVG_FORCE_ALLREDUCE") == "1":      # (one rank: the 1-rank RCCL group of the training leg's check)
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            backend = os.environ.get("DVG_DP_BACKEND", "nccl")
            # RCCL prints a version banner on STDOUT when its first communicator comes up: stdout belongs to the one JSON
            # line, so file descriptor 1 points at stderr until the communicator exists
            sys.stdout.flush()
            saved = os.dup(1)
            os.dup2(2, 1)
            try:
                if backend == "nccl":
                    dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.dev)
                else:
                    dist.init_process_group(backend, rank=self.rank, world_size=self.world)
                t = torch.ones(1, device=self.dev)
                dist.all_reduce(t)
                torch.cuda.synchronize()
                self.connected_ranks = int(t.item())
            finally:
                sys.stdout.flush()
                os.dup2(saved, 1)
                os.close(saved)
            self.backend = backend
            self.dist = dist
            # BEFORE anything is timed: the collective really spans --gpus ranks (an all-reduce of ones), else no number
            if self.connected_ranks != args.gpus or dist.get_world_size() != args.gpus:
                raise RuntimeError(f"bench.py: --gpus {args.gpus} but the {backend} group connects {self.connected_ranks} "
                                   f"rank(s) (world size {dist.get_world_size()})")

    def barrier(self):
        import torch
        if self.dist is not None:
            self.dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, dt: float) -> float:
        return max(self.all_ranks(dt))

    def all_ranks(self, dt: float) -> list:
        """Every rank's value of `dt`, in rank order, on every rank (a straggler is then visible in the JSON line)."""
        import torch
        if self.dist is None:
            return [dt]
        mine = torch.tensor([dt], device=self.dev, dtype=torch.float64)
        out = [torch.zeros_like(mine) for _ in range(self.world)]
        self.dist.all_gather(out, mine)
        return [float(t.item()) for t in out]


def make_gifs_leg(ctx: Ctx, args, model: str, nsample: int) -> dict:
    """BASELINE.json configs[2] (C3: "GP diverse sampling (generate_frames.py)"): `make_gifs` as generate_frames.py:107-189 runs it
    for one batch - the posterior rollout (:110-134), `nsample` sample rollouts with a GP draw at the trigger steps (:143-177;
    everything before the first predicted frame is the same for all samples of a batch and is computed once), utils.eval_seq's
    SSIM / PSNR of every predicted frame (:178) and best-of-N by mean SSIM (:188-189) - through generate_frames.Generator
    (rollout.GraphedSampler: the sample body as hipGraphs, `--inflight` samples at a time; the prediction steps before the first
    GP trigger step are sample-independent too and run once per batch - `per_sample_prediction` times the call without that).  KTH frames are 64 x 64 x 1 like
    Moving-MNIST (kth.py:54-55): same shapes, synthetic clips.  Predicted frames/s = B x n_future x nsample x ranks / wall time
    of the whole call (conditioning, posterior rollout and metrics included)."""
    import torch
    import generate_frames
    from dvg_amd.data import SyntheticMovingMNIST
    n_eval = args.n_past + args.n_future
    opt = generate_frames.build_parser().parse_args(["--synthetic_ckpt", "--batch_size", str(args.batch), "--model", model,
                                                     "--n_past", str(args.n_past), "--n_eval", str(n_eval),
                                                     "--inflight", str(max(1, args.inflight))])
    torch.manual_seed(args.seed + ctx.rank)
    g = generate_frames.Generator(opt, generate_frames.synthetic_checkpoint(opt), ctx.dev)
    x = SyntheticMovingMNIST(seq_len=n_eval, seed=args.seed + ctx.rank).batch_device(args.batch, ctx.dev)
    calibrate_batchnorm(g.encoder, g.decoder, x[0])
    reps = 2

    def timed():
        g.make_gifs(x, 3)           # warm-up: weight packs, the capture of the sample body
        ctx.barrier()
        t0 = time.perf_counter()
        for _ in range(reps):
            res = g.make_gifs(x, nsample)
        ctx.barrier()
        dt = ctx.max_over_ranks(time.perf_counter() - t0) / reps
        assert bool(torch.isfinite(res["ssim"]).all()) and bool(torch.isfinite(res["psnr"]).all())
        return dt, res

    dt, res = timed()
    first = [i for i in range(args.n_past, n_eval) if i % 15 == 0]
    shared = (min(first) if first else n_eval) - args.n_past
    # the same call with the reference loop's schedule: every sample runs all n_future prediction steps itself
    g.opt.no_share_prefix = True
    dt_ps, res_ps = timed()
    g.opt.no_share_prefix = False
    # (the GP draws of the two calls differ, so only the sample-independent part can be compared here)
    same = bool(torch.equal(res["posterior"], res_ps["posterior"])) and \
        bool(torch.equal(res["samples"][:, :args.n_past + shared], res_ps["samples"][:, :args.n_past + shared]))
    fps = lambda t: round(args.batch * args.n_future * nsample * ctx.world / t, 1)  # noqa: E731
    # frames the kernels actually produced per batch: the shared prediction steps once, the rest once per sample (ADVICE r04:
    # `predicted_frames_per_s` counts every DELIVERED sample frame, also those that are one computation shared by all samples)
    fps_computed = lambda t: round(args.batch * (shared + nsample * (args.n_future - shared)) * ctx.world / t, 1)  # noqa: E731
    return {"workload": f"make_gifs on one batch: posterior rollout + {nsample} sample rollouts (GP draw at i % 15 == 0) + SSIM / "
                        f"PSNR per predicted frame + best-of-N, {model}_64, batch {args.batch} per GPU, "
                        f"{args.n_past}-in/{args.n_future}-out, 64x64x1 synthetic clips (KTH-shaped)",
            "nsample": nsample, "samples_in_flight": max(1, args.inflight), "ms_per_batch": round(1e3 * dt, 2),
            "predicted_frames_per_s": fps(dt),
            "predicted_frames_per_s_is": "DELIVERED sample frames (B x n_future x nsample) / wall time; `computed_frames_per_s` "
                                         "counts the shared prediction steps once per batch",
            "computed_frames_per_s": fps_computed(dt),
            "schedule": f"conditioning and the {shared} prediction steps before the first GP trigger step run once per batch (they "
                        f"are the same kernels on the same inputs for every sample: bit-identical frames, tests/test_gpu_rollouts.py), "
                        f"the remaining {args.n_future - shared} steps once per sample",
            "per_sample_prediction": {"what": "the same call with every sample running all prediction steps itself (the "
                                              "reference loop's schedule, --no_share_prefix); conditioning still once per batch",
                                      "ms_per_batch": round(1e3 * dt_ps, 2), "predicted_frames_per_s": fps(dt_ps),
                                      "shared_part_bit_identical": same},
            "mean_best_ssim": round(float(res["ssim"].mean(2).max(1).values.mean()), 4)}


def hbm_bound_layers(ctx: Ctx, args) -> dict:
    """north_star's ">= 40 % HBM roofline on the encoder": the layers of the path whose roof IS the HBM (AI 4.4-19
    FLOP/B) are fused into or hidden behind other kernels in the timed rollout, so each is timed here in isolation at B = 64 -
    back-to-back launches between two HIP events on the launch stream, ALGORITHMIC bytes (operands + result, each once) over
    that time against the 8 TB/s peak.  first_conv: vgg_layer(1, 64) on the frame (vgg_64.py:23); last_projection: the decoder's
    ConvTranspose2d(64,1,3,1,1) + Sigmoid (vgg_64.py:88-92; projection + gather launches); lstm_step: embed + 2 cells + output
    (lstm.py:65-72; 3 launches, 4.4 MB of weights: latency-bound, reported in us); gp_sample: one sampling call of the GP
    trigger (gp_models.py:10-24)."""
    import torch
    from dvg_amd import fused, ops
    from dvg_amd.rollout import pooled_stream
    B = args.batch
    enc, dec, fp, gp, lik = build_models("vgg", B, 1, ctx.dev, args.seed)
    x = torch.rand(B, 1, 64, 64, device=ctx.dev)

    def timed(fn, reps=20, replays=10):
        """us per call: `reps` back-to-back calls captured as ONE hipGraph (no host launch latency between them - the rollout
        itself runs as graph replays), `replays` replays between two HIP events on the launch stream."""
        side = pooled_stream("warmup")
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            keep = [fn() for _ in range(reps)]
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(replays):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        del keep
        return e0.elapsed_time(e1) / (reps * replays) * 1e3

    def row(us, nbytes, launches, what):
        gbs = nbytes / (us * 1e-6) / 1e9
        return {"us": round(us, 2), "launches": launches, "algorithmic_bytes": int(nbytes), "gbs": round(gbs, 1),
                "frac_of_hbm_peak": round(gbs / PEAK_HBM_GBS, 4), "what": what}
    out = {}
    with torch.no_grad():
        c0 = enc.c1[0].main
        us = timed(lambda: fused.conv3_first_bn_act(c0[0], c0[1], x))
        out["first_conv"] = row(us, 4.0 * (x.numel() + B * 64 * 64 * 64 + 64 * 9), 1,
                                "conv_first_kernel<3,1>: (B,1,64,64) frame -> (B,64,64,64) NHWC, BN + LeakyReLU fused")
        d = ops.nhwc_empty(B, 64, 64, 64, ctx.dev).normal_()
        last = dec.upc5[1]
        us = timed(lambda: fused.convT3_last(last, d))
        out["last_projection"] = row(us, 4.0 * (d.numel() + B * 64 * 64 + 64 * 9), 2,
                                     "pixel_proj_kernel + convT_gather_kernel: (B,64,64,64) NHWC -> (B,1,64,64) frame, Sigmoid fused")
        h = torch.randn(B, 90, device=ctx.dev).tanh()
        h0 = fp.init_hidden()

        def lstm_step():
            fp.hidden = h0          # (the cells return new state tensors: h0 is never written)
            return fp(h)
        us = timed(lstm_step)
        nparam = sum(p.numel() for p in fp.parameters())
        out["lstm_step"] = row(us, 4.0 * (nparam + 6 * B * 256 + 2 * B * 90), 3,
                               "lstm_cell_x + lstm_cell + output GEMV: latency-bound (4.4 MB of weights from L2 / Infinity Cache)")
        eps = torch.randn(90, B, device=ctx.dev)

        def gp_sample():
            return lik(gp(h.transpose(0, 1).view(90, B, 1))).rsample(eps)
        gp_sample()
        us = timed(gp_sample, reps=5)
        out["gp_sample"] = row(us, 4.0 * (B * 90 * 4 + 90 * 40 * 43), 1,
                               "gp_predict_kernel (sampling): one 1024-thread workgroup per latent dim, fp64 inside: a serial "
                               "dependency chain in LDS, not a streaming kernel")
    return out


def gp_trigger_leg(ctx: Ctx, args, model: str, n_index: int = 8) -> dict:
    """generate_frames.py:249-298 (`--gp_trigger`, the other generate mode of BASELINE.json configs[2]) at the reference's own
    configuration: B = 50, 105 steps, per batch index.  ms per index of (a) the device schedule as generate_frames.py runs it
    (warm-up once per batch, decision and branch select on the device, the 93-step loop one hipGraph replay, logs read back
    once; an index whose decisions on an already computed trigger-free rollout are that rollout's decisions reuses it),
    (b) the same with every index running its own rollout, (c) the reference's schedule (`host_loop`: per index the warm-up,
    a `.cpu().numpy()` round trip and 2-3 encoder calls per step, eager launches: the r04 path)."""
    import torch
    import generate_frames
    from dvg_amd.data import SyntheticMovingMNIST
    B, total = 50, 105
    opt = generate_frames.build_parser().parse_args(["--synthetic_ckpt", "--batch_size", str(B), "--model", model,
                                                     "--n_eval", str(total)])
    torch.manual_seed(args.seed + ctx.rank)
    g = generate_frames.Generator(opt, generate_frames.synthetic_checkpoint(opt), ctx.dev)
    x = SyntheticMovingMNIST(seq_len=2, seed=args.seed + ctx.rank).batch_device(B, ctx.dev)
    calibrate_batchnorm(g.encoder, g.decoder, x[0])
    g.gp_trigger_gen(x, indices=[0], total=total)            # capture

    def timed(indices, **kw):
        ctx.barrier()
        t0 = time.perf_counter()
        res = g.gp_trigger_gen(x, indices=indices, total=total, **kw)
        ctx.barrier()
        return ctx.max_over_ranks(time.perf_counter() - t0) / len(indices), res
    dt, res = timed(list(range(n_index)))
    computed = g.trigger_rollouts_run
    dt_own, _ = timed([0, 1], share_paths=False)
    g.gp_trigger_gen(x, indices=[0], total=24, host_loop=True)      # warm the eager path
    dt_host, _ = timed([0], host_loop=True)
    assert all(bool(torch.isfinite(r["frames"]).all()) for r in res)
    return {"workload": f"GPtrigger_gen, {model}_64, batch {B}, {total} steps (12 warm-up + 93 with the variance-threshold "
                        "decision), per batch index", "indices_timed": n_index,
            "ms_per_index": round(1e3 * dt, 2), "rollouts_computed": computed,
            "ms_per_index_is": f"wall time of gp_trigger_gen over {n_index} batch indices / {n_index}: the batch's warm-up graph once, "
                               f"{computed} main-loop graph replay(s) - indices whose decisions on a computed trigger-free rollout "
                               "equal its decisions reuse its frames (their rollout is that rollout) -, one log read-back per index",
            "own_rollout_per_index": {"ms_per_index": round(1e3 * dt_own, 2),
                                      "batch_frames_per_s": round(B * total * ctx.world / dt_own, 1),
                                      "what": "share_paths=False: every index replays the main-loop graph"},
            "reference_schedule": {"what": "host_loop=True: the reference's statement order - warm-up per index, host round trip "
                                           "and 2-3 encoder calls per step, eager launches (the r04 path)",
                                   "ms_per_index": round(1e3 * dt_host, 2)},
            "speedup_over_reference_schedule": round(dt_host / dt, 2),
            "speedup_own_rollout_over_reference_schedule": round(dt_host / dt_own, 2),
            "triggers_index0": res[0]["triggers"][:12]}


# training shapes: (model, image width, channels, clips per GPU, n_past, n_future)
TRAIN_C4 = ("dcgan", 64, 3, 16, 2, 10)        # BASELINE.json configs[3]: BAIR 64x64 nc=3, batch 128 over 8 GPUs, 2-in/10-out
TRAIN_EXTRA = {                                # measured at N = 1 only (one hipGraph per iteration), reported under train.shapes
    "c2_vgg_64_b64": ("vgg", 64, 1, 64, 10, 10),      # configs[1]'s shape as a TRAINING iteration
    "c2_dcgan_64_b64": ("dcgan", 64, 1, 64, 10, 10),
    "c4_vgg_64": ("vgg", 64, 3, 16, 2, 10),
    "c5_vgg_128": ("vgg", 128, 3, 4, 4, 12),          # configs[4]: UCF 128x128, batch 32 over 8 GPUs, 4-in/12-out
    "c5_dcgan_128": ("dcgan", 128, 3, 4, 4, 12),
}


def measure_train(ctx: Ctx, args, graphed, allreduce: bool = True, shape=TRAIN_C4, iters=None) -> dict:
    """Data-parallel training at `shape` (default BASELINE.json configs[3]'s: BAIR-like dcgan_64, nc=3, 16 clips per GPU,
    2-in/10-out): train_model + both fine-tuning closures per iteration (train.py:354-361), gradients averaged over
    RCCL (dvg_amd/parallel.py).  Weak scaling: the global batch is clips per GPU x ranks."""
    import torch
    import train
    import utils
    from dvg_amd.data import SyntheticMovingMNIST, synthetic_video
    model, width, nc, per_gpu, n_past, n_future = shape
    iters = iters or args.train_iters
    T = n_past + n_future
    opt = train.build_parser().parse_args(["--model", model, "--channels", str(nc), "--image_width", str(width), "--dataset",
                                           "smmnist" if (nc, width) == (1, 64) else "bair",
                                           "--batch_size", str(per_gpu * ctx.world), "--n_past", str(n_past),
                                           "--n_future", str(n_future), "--no_save", "--synthetic_data"])
    opt.ft, opt.rank, opt.world, opt.local_batch = True, ctx.rank, ctx.world, per_gpu
    torch.manual_seed(args.seed)
    tr = train.Trainer(opt, ctx.dev)
    tr.train_mode()
    tr.set_allreduce(allreduce)
    if (nc, width) == (1, 64):
        seq = SyntheticMovingMNIST(seq_len=T, seed=args.seed + 31 * ctx.rank).batch(per_gpu)
    else:
        seq = synthetic_video(per_gpu, T, nc, width, seed=args.seed + 31 * ctx.rank)
    x, _ = utils.normalize_data(opt, torch.cuda.FloatTensor, seq)
    # graphed: False = eager, True = ONE hipGraph (collectives captured inside), "segmented" = a chain of hipGraphs cut at
    # the all-reduces, which stay eager (train.SegmentedIteration: what train.py runs with several ranks)
    step = (train.SegmentedIteration(tr, warmup=2) if graphed == "segmented" else
            train.GraphedIteration(tr, warmup=2) if graphed else tr.iteration)
    for _ in range(4 if graphed else 2):   # graphed: 2 eager warm-up iterations, the capture, one replay
        step(x)
    tr.reset_allreduce_stats()
    ctx.barrier()
    t0 = time.perf_counter()
    for _ in range(iters):
        step(x)
    ctx.barrier()
    per_rank = [t / iters for t in ctx.all_ranks(time.perf_counter() - t0)]
    dt = max(per_rank)
    assert all(bool(torch.isfinite(p).all()) for p in tr.encoder.parameters())
    res = {"ms_per_iter": round(1e3 * dt, 2), "per_rank_ms_per_iter": [round(1e3 * t, 2) for t in per_rank],
           "train_frames_per_s": round(per_gpu * ctx.world * (T - 1) / dt, 1)}
    st = tr.allreduce_stats()
    if st is not None and not graphed:
        res.update(st)
    if graphed == "segmented":
        res["graph_segments"] = step.n_segments
    return res


def train_leg(ctx: Ctx, args) -> dict:
    import torch
    out = {"config": "BAIR-shaped synthetic clips 64x64 nc=3, dcgan_64 + lstm + GP, 16 clips per GPU (global batch "
                     f"{16 * ctx.world}), 2-in/10-out, train_model + both fine-tuning closures per iteration",
           "parallelism": f"dp{ctx.world}: gradient all-reduce over RCCL, per-replica BatchNorm statistics",
           "scaling": "weak"}
    # proof that RCCL connected the ranks: the all-reduce of ones Ctx ran (and checked against --gpus) before any timing
    out["rccl_ranks"] = ctx.connected_ranks if ctx.dist is not None else 1
    out["eager"] = measure_train(ctx, args, graphed=False)
    if ctx.world > 1:
        out["eager_no_allreduce"] = measure_train(ctx, args, graphed=False, allreduce=False)
        out["allreduce_exposed_ms_per_iter"] = round(out["eager"]["ms_per_iter"] -
                                                     out["eager_no_allreduce"]["ms_per_iter"], 2)
    out["train_frames_per_s"] = out["eager"]["train_frames_per_s"]
    out["allreduce_ms_per_iter"] = out["eager"].get("allreduce_ms_per_iter")
    if ctx.world > 1 or (ctx.dist is not None and os.environ.get("DVG_FORCE_ALLREDUCE") == "1"):   # 2nd: one-rank RCCL check
        try:
            g = measure_train(ctx, args, graphed="segmented")
            out["hipgraph_segmented"] = g
            if g["train_frames_per_s"] > out["train_frames_per_s"]:
                out["train_frames_per_s"] = g["train_frames_per_s"]
                out["launch"] = "hipGraph segments, eager all-reduces between them"
        except Exception as e:   # noqa: BLE001 - reported, not fatal
            out["hipgraph_segmented"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


def graphed_train_leg(ctx: Ctx, args, result: dict, emit) -> None:
    """The same iteration replayed as ONE hipGraph (train.GraphedIteration); with ranks > 1 the graph contains the RCCL
    all-reduces (the chain-of-graphs form with eager all-reduces has been measured in train_leg by then).  Guarded: a
    watchdog emits the JSON line without this leg and ends the process if a captured collective hangs, and an exception
    is reported instead of raised."""
    collectives = ctx.world > 1 or (ctx.dist is not None and os.environ.get("DVG_FORCE_ALLREDUCE") == "1")
    if collectives and not args.train_full_graph:
        result["train"]["hipgraph"] = {"skipped": "collectives are never captured by default (c10d watchdog vs capturing "
                                                  "stream: intermittent process abort); see hipgraph_segmented, "
                                                  "--train-full-graph to try"}
        return
    done = threading.Event()

    def watchdog():
        if not done.wait(args.train_graph_timeout):
            result["train"]["hipgraph"] = {"error": f"no completion within {args.train_graph_timeout:.0f} s"}
            emit()
            # a hung captured collective cannot be recovered from inside the process: the headline line is out, now FAIL -
            # the launcher (torch.distributed.run / self_launch) sees a non-zero exit and tears the other ranks down.
            # os._exit: no atexit / destructor may touch the wedged GPU queue; never re-exec.
            os._exit(3)
    if collectives:
        threading.Thread(target=watchdog, daemon=True).start()
    try:
        g = measure_train(ctx, args, graphed=True)
        result["train"]["hipgraph"] = g
        if g["train_frames_per_s"] > result["train"]["train_frames_per_s"]:
            result["train"]["train_frames_per_s"] = g["train_frames_per_s"]
            result["train"]["launch"] = "hipGraph replay"
    except Exception as e:   # noqa: BLE001 - reported, not fatal: the headline metric has been measured
        result["train"]["hipgraph"] = {"error": f"{type(e).__name__}: {e}"[:300]}
    done.set()
    if ctx.world == 1 and not collectives and not args.no_train_shapes:
        # the other BASELINE training shapes on one GPU, each as one hipGraph per iteration (per-GPU shapes of C4 / C5; the
        # C2 shape trained): train frames/s = clips x (n_past + n_future - 1) / iteration
        import torch
        shapes = {}
        for name, shape in TRAIN_EXTRA.items():
            try:
                g = measure_train(ctx, args, graphed=True, shape=shape, iters=3)
                shapes[name] = {"shape": "%s_%d nc=%d, %d clips, %d-in/%d-out" % shape, "ms_per_iter": g["ms_per_iter"],
                                "train_frames_per_s": g["train_frames_per_s"]}
            except Exception as e:   # noqa: BLE001
                shapes[name] = {"error": f"{type(e).__name__}: {e}"[:200]}
            torch.cuda.empty_cache()
        result["train"]["shapes"] = shapes
