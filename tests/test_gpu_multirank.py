"""The multi-rank control flow of bench.py, rehearsed on the ONE GPU of the test box: `python bench.py --gpus 2` from a bare
environment must start its own launcher as a child process, both ranks must run the replica rollouts and the data-parallel
training leg (three gradient all-reduces per iteration over the flat arena) and rank 0 must print one JSON line.  RCCL refuses
two ranks on one device, so the rehearsal switches put both ranks on GPU 0 with the gloo backend (DVG_DP_SHARE_GPU=1,
DVG_DP_BACKEND=gloo): the numbers mean nothing and the line says so; the code path - self-launch, rendezvous, barriers,
max-over-ranks timing, ArenaReducer ranges, the guarded graphed leg - is the one the 8-GPU run takes."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port() -> int:
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("ranks", [2, 4])
def test_bench_ranks_rehearsal(ranks):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DVG_DP_SHARE_GPU="1", DVG_DP_BACKEND="gloo", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "3", "--warmup", "1",
                        "--model", "dcgan", "--no-families", "--no-cpu-baseline", "--train-iters", "1",
                        "--train-graph-timeout", "120"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == ranks and d["scaling"] == "weak" and "rehearsal" in d
    assert d["value"] > 0 and d["config"]["rollouts_in_flight"] == 3
    assert len(d["per_rank_ms_per_step"]) == ranks and max(d["per_rank_ms_per_step"]) == d["ms_per_step"]
    t = d["train"]
    assert t["rccl_ranks"] == ranks                   # from an actual all-reduce of ones, checked against --gpus before timing
    assert len(t["eager"]["per_rank_ms_per_iter"]) == ranks
    assert t["eager"]["allreduces_per_iter"] == 3.0   # decoder / LSTM / GP range, encoder range, [GP | LSTM] of both fine-tuning closures
    assert t["eager"]["allreduce_MB_per_iter"] > 40
    assert "eager_no_allreduce" in t and "hipgraph" in t
    assert t["hipgraph_segmented"]["graph_segments"] == 4, t["hipgraph_segmented"]   # cut at the three all-reduce groups


def test_bench_refuses_a_group_that_does_not_span_gpus_ranks():
    """`--gpus 2` under a launcher that only started ONE rank: WORLD_SIZE disagrees -> exit code 2 before any GPU work."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and not r.stdout.strip()


@pytest.mark.parametrize("ranks", [2, 4])
def test_train_py_ranks_end_with_identical_parameters(ranks):
    """train.py under `torch.distributed.run` with two ranks (rehearsal switches: both on GPU 0, gloo): identical initial
    parameters on both ranks, different data per rank, gradients averaged over the flat arena in place, the iteration
    replayed as a chain of hipGraphs with the all-reduces eager between them (train.SegmentedIteration) - after four
    iterations (train_model + both fine-tuning closures each) every rank must hold bit-identical parameters, and they must
    differ from a single-rank run on rank 0's data alone (i.e. the other rank's gradients did arrive)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DVG_DP_SHARE_GPU="1", DVG_DP_BACKEND="gloo", OMP_NUM_THREADS="2")
    args = ["--model", "dcgan", "--dataset", "smmnist", "--n_past", "2", "--n_future", "3", "--n_eval", "5", "--niter", "1",
            "--epoch_size", "4", "--no_save", "--save_every", "1000", "--print_param_checksum"]

    def run(cmd):
        r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return {ln.split()[1]: ln.split()[-2:] for ln in r.stdout.splitlines() if "param checksum" in ln}

    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranks}", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "train.py"), "--batch_size",
               str(4 * ranks)] + args)
    assert set(two) == {str(r) for r in range(ranks)}, two
    assert all(two[str(r)] == two["0"] for r in range(ranks)), two
    one = run([sys.executable, os.path.join(ROOT, "train.py"), "--batch_size", "4"] + args)
    assert one["0"] != two["0"]


def test_library_loaded_before_torch_touches_the_gpu_still_launches():
    """build() followed by smoke() in ONE process loads libdvg_hip.so before torch has initialised its (bundled) HIP
    runtime; the library links the system runtime, and in that order every launch used to fail with "no ROCm-capable
    device is detected".  `_lib.lib()` now brings torch's runtime up first."""
    code = ("from dvg_amd import _lib\n"
            "assert _lib.lib().dvg_abi_version() == 9\n"
            "import torch\n"
            "from dvg_amd import ops\n"
            "x = torch.arange(2 * 3 * 4 * 5, dtype=torch.float32, device='cuda').reshape(2, 3, 4, 5)\n"
            "y = ops.to_nhwc(x)\n"
            "torch.cuda.synchronize()\n"
            "assert torch.equal(y, x) and ops.is_nhwc(y)\n"
            "print('launch ok')\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=300)
    assert r.returncode == 0 and "launch ok" in r.stdout, r.stderr[-1500:]


@pytest.mark.parametrize("model,linear", [("dcgan", False), ("vgg", False), ("vgg", True)])
def test_two_ranks_with_sync_bn_train_like_one_process(model, linear, tmp_path):
    """VERDICT r05 item 5 / SURVEY 8(e): the reference is one process with full-batch BatchNorm statistics.  2 ranks x B/2
    clips with `--sync_bn` (BatchNorm sums all-reduced forward and backward, gradients averaged over the arena) must train
    like 1 rank x B clips on the same global batches (tools/dp_equivalence.py; rehearsal switches: both ranks on GPU 0,
    gloo); per-replica statistics (the default, DDP semantics) must not - the control that shows the comparison can fail.
      * FORWARD, every case: BatchNorm running statistics after the first pass to 1e-4 (measured 9e-7 ... 2.6e-5), train_model's
        loss values (mean over the ranks) in the first stepping iteration - every run starts it from identical parameters -
        to 1e-5 (measured 3e-8 ... 1.5e-7);
      * the averaged GRADIENTS of the first train_model backward.  Two forward passes that agree to rounding still decide the
        odd LeakyReLU / max-pool BRANCH differently (a pre-activation within ~1e-7 of zero; a handful among the 4e7 of a
        vgg_64 pass at this batch), and the backward pass through BatchNorm at batch 8 amplifies that seed ~1.5 x per layer.
        dcgan_64 (10 layers, 6e6 pre-activations: no flip in this configuration): EVERY parameter tensor to 1e-4 max-norm,
        the typical one (median) to 1e-5 - measured <= 4e-5 (an LSTM weight whose gradient is 3e-7) / 5e-7.  vgg_64 with every
        LeakyReLU made linear (`--linear_lrelu`, all three runs): the whole decoder - 13 BatchNorm layers deep -, the LSTM and
        the GP to 1e-4 (measured <= 9e-6), the encoder below its max-pools to 1e-2 in relative L2.  vgg_64 as it is: every
        tensor to 2e-2 in relative L2 (measured 3.5e-3) against the control's ~1;
      * parameters right after that closure's Adam steps: Adam's first step is sign-like (m / sqrt(v) = +-1 whatever |g| is), so
        an entry whose gradient is at rounding level may step the other way by 2 lr - the fraction of entries that differ by
        more than 1e-5 of the tensor's largest magnitude is bounded (dcgan_64: < 5e-3, measured 3e-4; the control: 0.25);
      * after three iterations: relative L2 distance per module < 3e-2 and far below the control's."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(DVG_DP_SHARE_GPU="1", DVG_DP_BACKEND="gloo", OMP_NUM_THREADS="2")
    # (Until the last day of r06 the vgg_64 case ran with DVG_WINOGRAD=0: with a second process on the device the F(4x4) weight
    # transform wrote zero rows into U in 1-30 % of its launches - winograd.hip, wrow_owner_note; test_packed_weights_with_a_second_process
    # below guards the fix.)
    script = os.path.join(ROOT, "tools", "dp_equivalence.py")
    common = ["--model", model, "--batch", "8", "--iters", "3"] + (["--linear_lrelu"] if linear else [])

    def run(world, extra, name):
        out = str(tmp_path / name)
        cmd = [sys.executable] + ([] if world == 1 else ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                                                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port())])
        r = subprocess.run(cmd + [script] + common + extra + ["--out", out], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-2000:]
        return torch.load(out)
    one = run(1, [], "one.pt")
    sync = run(2, ["--sync_bn"], "sync.pt")
    plain = run(2, [], "plain.pt")
    assert sync["sync_bn"] and sync["world"] == 2 and not plain["sync_bn"] and one["world"] == 1

    def per_tensor(a, b):
        out = {}
        for k in b:
            den = float(b[k].abs().max())
            if den == 0.0:
                assert float((a[k] - b[k]).abs().max()) == 0.0, k
                continue
            out[k] = (float((a[k] - b[k]).abs().max()) / den, float((a[k] - b[k]).double().norm() / b[k].double().norm()))
        return out

    def l2(a, b, prefix):
        num = sum(float(((a[k] - b[k]).double() ** 2).sum()) for k in b if k.startswith(prefix))
        den = sum(float((b[k].double() ** 2).sum()) for k in b if k.startswith(prefix))
        return (num / max(den, 1e-300)) ** 0.5

    def beyond(a, b):
        n = bad = 0
        for k in b:
            tol = 1e-5 * max(float(b[k].abs().max()), 1e-12)
            bad += int(((a[k] - b[k]).abs() > tol).sum())
            n += b[k].numel()
        return bad / n

    def figures(run_, tag):
        g = per_tensor(run_["grads_first_backward"], one["grads_first_backward"])
        mx = sorted(v[0] for v in g.values())
        wk = max(g, key=lambda k: g[k][0])
        upper = [v[0] for k, v in g.items() if not k.startswith("encoder.c") or k.startswith("encoder.c5")]
        res = {"worst": g[wk][0], "median": mx[len(mx) // 2], "worst_l2": max(v[1] for v in g.values()),
               "upper_worst": max(upper), "enc_l2": max(v[1] for k, v in g.items() if k.startswith("encoder."))}
        b = max(per_tensor(run_["buffers_first_forward"], one["buffers_first_forward"]).values())[0]
        # (mse_latent, loss) of train_model in the first stepping iteration: computed BEFORE any parameter moved
        loss1 = max(abs(u - v) / max(abs(v), 1e-12) for u, v in zip(run_["losses"][0][:2], one["losses"][0][:2]))
        f1 = beyond(run_["params_first_step"], one["params_first_step"])
        mods = ("encoder", "decoder", "frame_predictor", "gp_layer", "likelihood")
        pend = {m: float(f"{l2(run_['params'], one['params'], m):.1e}") for m in mods}
        print(f"   {tag}: gradients of the first backward - worst tensor {res['worst']:.2e} ({wk}), median {res['median']:.2e}, worst "
              f"relative L2 {res['worst_l2']:.2e}; decoder / LSTM / GP / encoder head worst {res['upper_worst']:.2e}, encoder relative L2 "
              f"{res['enc_l2']:.2e} | BatchNorm buffers {b:.2e} | train_model losses {loss1:.2e} | entries beyond 1e-5 after the first "
              f"Adam step {f1:.2e} | rel-L2 of the parameters after 3 iterations {pend}")
        return res, b, loss1, f1, pend
    print(f"\ndp equivalence {model}_64{' (LeakyReLU linear)' if linear else ''}: 2 ranks x 4 clips against 1 rank x 8 clips")
    g_s, b_s, l_s, f_s, p_s = figures(sync, "sync-BN")
    g_p, b_p, l_p, f_p, p_p = figures(plain, "per-replica BatchNorm (control)")
    assert b_s < 1e-4 and l_s < 1e-5, (b_s, l_s)
    if model == "dcgan":
        assert g_s["worst"] < 1e-4 and g_s["median"] < 1e-5 and f_s < 5e-3, (g_s, f_s)
    elif linear:
        assert g_s["upper_worst"] < 1e-4 and g_s["enc_l2"] < 1e-2, g_s
    else:
        assert g_s["worst_l2"] < 2e-2, g_s
    assert max(p_s.values()) < 3e-2, p_s
    # the control: per-replica statistics are a different computation
    assert g_p["median"] > 100 * g_s["median"] and g_p["worst_l2"] > 20 * g_s["worst_l2"] and b_p > 1e-3, (g_p, b_p)
    assert f_p > 5 * f_s and p_p["encoder"] > 3 * p_s["encoder"], (f_p, f_s, p_p, p_s)


def test_packed_weights_with_a_second_process_on_the_device():
    """The per-weight-version cache entries of the training path (packed igemm weights, Winograd-domain weights of the forward
    and of the data gradient, transposes) recomputed while ANOTHER PROCESS trains on the same device must come out bit-identical
    every time (tools/diag_pack_repeat.py).  r06: in its earlier form the F(4x4) weight transform wrote whole rows of U as zeros in
    1-30 % of its launches under exactly this contention (193 of 5 700 recomputations; 0 alone on the device; a wrong packed-FMA result) - the cause of the one-GPU
    rehearsal's run-to-run different vgg_64 gradients (profiles/r06_dp_race_bisect.txt)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "diag_pack_repeat.py"), "--noise", "train", "--iters", "40"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "diag_pack_repeat: noise train: 0 differing recomputations" in r.stdout, r.stdout[-2000:]
