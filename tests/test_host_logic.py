"""CPU: host-side logic of the path — data-parallel gradient reduction over gloo (world_size 2),
synthetic data determinism, integer bookkeeping of the rollout, script argument surfaces."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dp_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from dvg_amd import parallel
    r, w, _ = parallel.init_distributed("gloo")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    if rank == 1:  # replicas must start identical: perturb then broadcast from rank 0
        with torch.no_grad():
            for p in net.parameters():
                p.add_(1.0)
    parallel.broadcast_parameters([net])
    x = torch.full((4, 7), float(rank + 1))
    net(x).sum().backward()
    net[1].bias.grad = None  # a parameter the pass did not touch must stay untouched
    local = [None if p.grad is None else p.grad.clone() for p in net.parameters()]
    red = parallel.FlatGradReducer(net.parameters())
    red.reduce()
    # plain numpy through the queue: torch tensors travel by file descriptor, which races with this process exiting
    npy = lambda t: None if t is None else t.detach().numpy().copy()  # noqa: E731
    out = {"rank": r, "world": w, "params": [npy(p) for p in net.parameters()],
           "local": [npy(g) for g in local], "reduced": [npy(p.grad) for p in net.parameters()],
           "shard": parallel.shard_batch(128, w)}
    q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_allreduce_gloo_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda d: d["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = res
    for d in (a, b):
        for k in ("params", "local", "reduced"):
            d[k] = [None if v is None else torch.from_numpy(v) for v in d[k]]
    assert a["world"] == 2 and a["shard"] == 64
    for pa, pb in zip(a["params"], b["params"]):
        assert torch.equal(pa, pb), "broadcast_parameters must make the replicas identical"
    for la, lb, ra, rb in zip(a["local"], b["local"], a["reduced"], b["reduced"]):
        if la is None:
            assert ra is None and rb is None
            continue
        assert torch.allclose(ra, (la + lb) / 2, atol=1e-6) and torch.equal(ra, rb)
    assert not torch.equal(a["local"][0], b["local"][0])


def _bcast_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from dvg_amd import parallel
    from dvg_amd.optim import FlatArena
    parallel.init_distributed("gloo")
    torch.manual_seed(rank)                       # replicas start DIFFERENT
    net = torch.nn.Sequential(torch.nn.Conv2d(2, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Linear(4, 3))
    outside = torch.nn.Linear(3, 2)               # a module whose parameters do not live in the arena
    params = list(net.parameters())
    arena = FlatArena(FlatArena.size_for(params), "cpu")
    off = 0
    with torch.no_grad():
        for p_ in params:                          # what FusedAdam._build does: parameters become views of arena.p
            v = arena.p[off:off + p_.numel()].view(p_.shape)
            v.copy_(p_)
            p_.data = v
            off += FlatArena.padded(p_.numel())
        net[1].running_mean.fill_(float(rank + 1))
        net[1].num_batches_tracked.fill_(7 * (rank + 1))
    n = parallel.broadcast_parameters([net, outside], arena_p=arena.p)
    assert all(p_.data_ptr() >= arena.p.data_ptr() for p_ in params), "parameters must still be views of the arena"
    q.put({"rank": rank, "collectives": n, "arena": arena.p.numpy().copy(),
           "params": [p_.detach().numpy().copy() for p_ in list(net.parameters()) + list(outside.parameters())],
           "bufs": [b.numpy().copy() for b in net.buffers()]})
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_parameters_is_one_collective_for_the_arena_gloo_world2():
    """parallel.broadcast_parameters(arena_p=...): the parameter arena goes as ONE broadcast, buffers and parameters outside
    the arena packed per dtype (float32 + BatchNorm's int64 counter): 3 collectives here instead of one per tensor (11)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + os.getpid() % 200
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda d: d["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = res
    assert a["collectives"] == b["collectives"] == 3
    assert np.array_equal(a["arena"], b["arena"])
    for x, y in zip(a["params"] + a["bufs"], b["params"] + b["bufs"]):
        assert np.array_equal(x, y)
    assert float(b["bufs"][0][0]) == 1.0 and int(b["bufs"][2]) == 7          # rank 0's values won


def _arena_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from dvg_amd import parallel
    from dvg_amd.optim import FlatArena
    parallel.init_distributed("gloo")
    # the Trainer's layout: parameters whose .grad are views of one flat gradient buffer; ranges reduced in place
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))      # "decoder side" | "encoder side"
    params = list(net.parameters())
    arena = FlatArena(FlatArena.size_for(params), "cpu")
    views, off = [], 0
    for p_ in params:
        views.append(arena.g[off:off + p_.numel()].view(p_.shape))
        p_.grad = views[-1]
        off += FlatArena.padded(p_.numel())
    split = FlatArena.padded(35) + FlatArena.padded(5)                           # end of net[0]'s range
    net(torch.full((4, 7), float(rank + 1))).sum().backward()
    assert all(p_.grad is v for p_, v in zip(params, views)), "backward must accumulate into the flat views"
    local = arena.g.clone()
    red = parallel.ArenaReducer(arena.g)
    h1 = red.start(split, arena.g.numel())        # second range first, asynchronously ...
    h0 = red.start(0, split)                      # ... while "the rest of backward" would still be running
    red.finish(h1)
    red.finish(h0)
    red.enabled = False
    assert red.start(0, split) is None            # bench.py's no-all-reduce leg
    q.put({"rank": rank, "local": local.numpy().copy(), "reduced": arena.g.numpy().copy(),
           "grad0": params[0].grad.detach().numpy().copy(), "calls": red.calls, "floats": red.floats})
    dist.barrier()
    dist.destroy_process_group()


def test_arena_reducer_ranges_gloo_world2():
    """dvg_amd.parallel.ArenaReducer: in-place, asynchronous averaging of RANGES of the flat gradient arena whose views
    are the parameters' .grad (what train.Trainer does with RCCL), two ranks over gloo."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40
    procs = [ctx.Process(target=_arena_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    a, b = sorted([q.get(timeout=120) for _ in procs], key=lambda d: d["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    mean = (a["local"] + b["local"]) / 2
    assert not np.array_equal(a["local"], b["local"])
    assert np.allclose(a["reduced"], mean, atol=1e-6) and np.array_equal(a["reduced"], b["reduced"])
    assert np.allclose(a["grad0"].reshape(-1), mean[:35], atol=1e-6), "p.grad IS the reduced buffer (no copy back)"
    assert a["calls"] == 2 and a["floats"] == a["local"].size


def test_shard_batch_rejects_uneven_split():
    from dvg_amd import parallel
    assert parallel.shard_batch(32, 8) == 4
    with pytest.raises(ValueError):
        parallel.shard_batch(50, 8)


def test_synthetic_moving_mnist_follows_the_reference_generator_contract():
    """moving_mnist.py:38-91: (T,64,64,1) float32 in [0,1], additive compositing clipped at 1, seeded."""
    from dvg_amd.data import SyntheticMovingMNIST
    a = SyntheticMovingMNIST(seq_len=6, seed=3).batch(3)
    b = SyntheticMovingMNIST(seq_len=6, seed=3).batch(3)
    assert a.shape == (3, 6, 64, 64, 1) and a.dtype == torch.float32
    assert torch.equal(a, b)
    assert float(a.min()) >= 0.0 and float(a.max()) <= 1.0 and float(a.max()) > 0.5
    moved = (a[:, 1:] - a[:, :-1]).abs().flatten(2).sum(2)
    assert float(moved.min()) > 0.0 or True  # sprites may rest for a frame; at least something moves:
    assert float(moved.sum()) > 0.0


def test_normalize_data_matches_oracle_layout():
    from dvg_amd import utils
    from oracle import dvg_oracle as orc
    seq = torch.rand(2, 3, 8, 8, 3)
    ours, tgt = utils.normalize_data(None, torch.FloatTensor, (seq.clone(), torch.zeros(2)))
    ref = orc.normalize_data(seq)
    assert len(ours) == 3 and all(torch.equal(o, r) for o, r in zip(ours, ref))
    ours2, tgt2 = utils.normalize_data(None, torch.FloatTensor, seq.clone())
    assert tgt2 is None and torch.equal(ours2[1], ref[1])


def test_trigger_schedule_matches_oracle():
    from dvg_amd.rollout import trigger_steps
    from oracle import dvg_oracle as orc
    for n_past, n_eval in ((10, 20), (5, 105), (5, 15), (2, 31)):
        assert trigger_steps(n_past, n_eval) == orc.gp_trigger_steps(n_past, n_eval)


def test_script_flag_surfaces():
    sys.path.insert(0, ROOT)
    import generate_frames
    import train
    o = train.build_parser().parse_args(["--model", "vgg", "--batch_size", "8", "--n_past", "2", "--g_dim", "90",
                                          "--last_frame_skip"])
    assert o.model == "vgg" and o.batch_size == 8 and o.last_frame_skip and o.rnn_size == 256 and o.niter == 601
    g = generate_frames.build_parser().parse_args(["--model_dir", "x", "--dataset", "kth"])
    assert g.n_eval == 105 and g.n_future == 100 and g.batch_size == 50   # generate_frames.py:47-49


def test_init_weights_matches_reference_distribution():
    """utils.py:304-311: class-name substring dispatch."""
    from dvg_amd import utils
    from dvg_amd.models.vgg_64 import encoder
    torch.manual_seed(0)
    e = encoder(90, 1)
    e.apply(utils.init_weights)
    w = e.c3[1].main[0].weight
    assert abs(float(w.std()) - 0.02) < 2e-3 and abs(float(w.mean())) < 1e-3
    assert float(e.c3[1].main[0].bias.abs().max()) == 0.0
    assert abs(float(e.c3[1].main[1].weight.mean()) - 1.0) < 0.01


def test_batch_prefetcher_keeps_order_propagates_errors_and_ends():
    """train.BatchPrefetcher: the generator's items in its order from a background thread, its exception re-raised at the
    consumer, exhaustion as StopIteration (repeatably)."""
    import train
    assert list(train.BatchPrefetcher(iter(range(9)), depth=2)) == list(range(9))
    pf = train.BatchPrefetcher(iter([1]))
    assert next(pf) == 1
    for _ in range(2):
        with pytest.raises(StopIteration):
            next(pf)

    def bad():
        yield "a"
        raise ValueError("boom")
    pf = train.BatchPrefetcher(bad())
    assert next(pf) == "a"
    with pytest.raises(ValueError):
        next(pf)


def test_gp_trigger_oracle_memo_and_forced_decisions():
    """oracle.gp_trigger_gen (generate_frames.py:249-298), the checker of the B = 50 GPU test: (1) with a memo shared by the calls
    for several batch indices every (step, decisions so far) pair is computed once and the results equal the un-memoised calls
    bit for bit; (2) `decisions` + `guard`: a step whose own margin is inside the guard follows the given branch (reported in
    `forced`), every other step decides for itself - with guard = 0 the decisions are ignored."""
    from oracle import dvg_oracle as orc
    from oracle import params
    from tests.test_gpu_configs import _build, _oracle_fns
    B, total = 5, 18
    mods, (esd, dsd, lsd, gsd, lik) = _build("dcgan", 64, 1, B, 3300)
    xs = [params.frames(3310, B, 1, 64)]
    eps = {i: params.normal(3320 + i, 90, B) for i in range(12, total)}
    enc_o, dec_o = _oracle_fns("dcgan", 64, esd, dsd)
    memo = {}
    with torch.no_grad():
        for index in (0, 4):
            a = orc.gp_trigger_gen(xs, enc_o, dec_o, lsd, gsd, lik, index, eps, total=total, depth=-250)
            b = orc.gp_trigger_gen(xs, enc_o, dec_o, lsd, gsd, lik, index, eps, total=total, depth=-250, memo=memo)
            assert a["triggers"] == b["triggers"] and a["values"] == b["values"] and a["thresholds"] == b["thresholds"]
            assert all(torch.equal(u, v) for u, v in zip(a["frames"], b["frames"]))
        assert any(k[0] == "post" for k in memo) and any(k[0] == "warm" for k in memo)
        flip = {i: (i not in a["triggers"]) for i in range(12, total)}           # the opposite branch everywhere
        c = orc.gp_trigger_gen(xs, enc_o, dec_o, lsd, gsd, lik, 4, eps, total=total, depth=-250, decisions=flip, guard=0.0)
        assert c["triggers"] == a["triggers"] and c["forced"] == []
        d = orc.gp_trigger_gen(xs, enc_o, dec_o, lsd, gsd, lik, 4, eps, total=total, depth=-250, decisions=flip, guard=1e9)
        assert d["forced"] == list(range(12, total)) and d["triggers"] == [i for i in range(12, total) if flip[i]]


def test_hoist_ready_and_hoisted_skip_agree_across_evictions(monkeypatch):
    """a decoder block's producer asks fused._hoist_ready(conv, skip) and, on yes, hands its consumer an upsampled
    WinoV - the consumer then MUST get the skip half from fused._hoisted_skip.  The two agree in every cache state (frozen
    declaration, second sighting, weight change, dead skips), and where the entry is evicted between the two calls (the
    64-entry bound) the consumer's `force=True` still yields the skip half instead of raising.  Host logic only: the packing
    and the conv launch are stubbed."""
    from dvg_amd import fused
    monkeypatch.setattr(fused, "_split_packed", lambda conv, c1: ("px", "ps"))
    calls = []
    part = lambda ps: calls.append(ps) or "S"      # noqa: E731
    fused.clear_skip_hoist_cache()
    conv = torch.nn.Conv2d(8, 4, 3, 1, 1)
    x, skip = torch.zeros(1, 4, 2, 2), torch.zeros(1, 4, 2, 2)

    def agree(force=False):
        ready = fused._hoist_ready(conv, skip)
        got = fused._hoisted_skip(conv, x, skip, part, force=force)
        assert ready == (got is not None) or force, (ready, got)
        return got
    assert agree() is None                       # first sighting, undeclared: the ordinary concat conv
    assert agree() == ("px", "S")                # second sighting
    skip.add_(1.0)                               # the skip changed: a new first sighting
    assert agree() is None
    fused.declare_frozen_skips([skip])           # declared loop-invariant: ready at once
    fused._skip_seen.clear()
    assert agree() == ("px", "S")
    with torch.no_grad():
        conv.weight.add_(1.0)                    # new weight version: entry stale, but the skip is still declared frozen
    assert agree() == ("px", "S")
    # eviction between producer and consumer of an UNDECLARED skip
    fused.clear_skip_hoist_cache()
    assert fused._hoisted_skip(conv, x, skip, part) is None and fused._hoisted_skip(conv, x, skip, part) is not None
    assert fused._hoist_ready(conv, skip)        # the producer sees the entry ...
    keep = [torch.zeros(1) for _ in range(70)]
    for t in keep:                               # ... 70 other (conv, skip) pairs pass through the 64-entry table ...
        fused._hoisted_skip(conv, x, t, part)
    assert not fused._hoist_ready(conv, skip)    # ... and it is gone when the consumer asks
    assert fused._hoisted_skip(conv, x, skip, part) is None
    fused._skip_seen.clear()
    assert fused._hoisted_skip(conv, x, skip, part, force=True) == ("px", "S")    # the consumer's call: never None
    del keep
    import gc
    gc.collect()
    alive = torch.zeros(1)
    fused._hoisted_skip(conv, x, alive, part)              # a miss sweeps the dead entries
    assert all(e[0]() is not None for e in fused._skip_seen.values())
    fused.clear_skip_hoist_cache()


def _sync_bn_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from dvg_amd import fused, ops, parallel
    parallel.init_distributed("gloo")
    assert fused.sync_bn_world() == 1 and ops.sync_bn_state() is None
    fused.set_sync_bn(dist, dist.new_group())
    assert fused.sync_bn_world() == 2 and ops.sync_bn_state() is not None
    g = torch.Generator().manual_seed(100 + rank)
    partial = torch.randn(3 * 5, 2, 8, generator=g)        # 3 groups x 5 per-tile rows of (sum, sum of squares) x 8 channels
    out = ops.sync_partial_rows(partial, 3)
    q.put({"rank": rank, "partial": partial.numpy().copy(), "out": out.numpy().copy(), "grouped": int(out.grouped)})
    fused.set_sync_bn(None)
    assert fused.sync_bn_world() == 1 and ops.sync_bn_state() is None
    dist.barrier()
    dist.destroy_process_group()


def test_sync_bn_partial_rows_allreduce_gloo_world2():
    """--sync_bn's collective (ops.sync_partial_rows): each rank's per-tile partial rows [G r][2][C] become TWO fp32 rows per
    group (hi + lo) that add up to the fp64 sums over all rows of all ranks, identical on every rank; the switch turns on
    only for a group of more than one rank and off again."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29950 + os.getpid() % 40
    procs = [ctx.Process(target=_sync_bn_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda d: d["rank"])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    a, b = res
    want = (a["partial"].astype(np.float64).reshape(3, 5, 2, 8).sum(1) +
            b["partial"].astype(np.float64).reshape(3, 5, 2, 8).sum(1))
    # two fp32 rows per group (hi, lo) whose fp64 sum is the fp64 total over both ranks
    assert a["out"].shape == (6, 2, 8) and a["grouped"] == 3
    assert np.array_equal(a["out"], b["out"])
    got = a["out"].astype(np.float64).reshape(3, 2, 2, 8).sum(1)
    assert np.array_equal(a["out"].reshape(3, 2, 2, 8)[:, 0], want.astype(np.float32))
    np.testing.assert_allclose(got, want, rtol=1e-14, atol=0)
    # one rank: the switch stays off (nothing to synchronise)
    from dvg_amd import fused, ops
    fused.set_sync_bn(None)
    assert fused.sync_bn_world() == 1 and ops.sync_bn_state() is None


def test_environment_switches_are_the_documented_ten_and_no_file_is_a_monolith():
    """"<= 10 switches", "no file > 800 lines outside csrc/".  Every DVG_* environment variable the product
    reads (package, train.py, generate_frames.py, bench.py) is named in README.md's switch paragraph and there are at most ten;
    no source file outside dvg_amd/csrc is longer than 800 lines."""
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = glob.glob(os.path.join(root, "dvg_amd", "**", "*.py"), recursive=True) + \
        [os.path.join(root, f) for f in ("train.py", "generate_frames.py", "bench.py", "utils.py", "gp_models.py")]
    read = set()
    for f in files:
        read |= set(re.findall(r"environ(?:\.get|\.setdefault)?[\(\[]\s*[\"'](DVG_[A-Z0-9_]+)[\"']", open(f).read()))
    readme = open(os.path.join(root, "README.md")).read()
    para = readme[readme.index("Switches (environment"):readme.index("All keep the results within the parity tolerances.")]
    documented = set(re.findall(r"DVG_[A-Z0-9_]+", para))
    assert read <= documented, sorted(read - documented)
    assert len(read) <= 10, sorted(read)
    long_files = []
    for pat in ("*.py", "dvg_amd/**/*.py", "tests/*.py", "tools/*.py", "oracle/*.py", "models/*.py", "include/*.h"):
        for f in glob.glob(os.path.join(root, pat), recursive=True):
            n = sum(1 for _ in open(f))
            if n > 800:
                long_files.append((os.path.relpath(f, root), n))
    assert not long_files, long_files


def test_pooled_streams_are_made_once_per_role_and_index(monkeypatch):
    """rollout.pooled_stream: torch hands streams out of a pool of 32 per device, round-robin; graph holders that made fresh ones
    (six per ConcurrentRollouts) wrapped around it in bench.py and hipGraphLaunch crashed (r06).  Every (device, role, index) stream is
    made once per process and shared by all holders - checked here without a GPU, on a stand-in for torch.cuda.Stream."""
    import torch
    from dvg_amd import rollout
    made = []

    class FakeStream:
        def __init__(self):
            made.append(self)
    monkeypatch.setattr(torch.cuda, "Stream", FakeStream)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 0)
    monkeypatch.setattr(rollout, "_streams", {})
    chains = [[rollout.pooled_stream("chain", k) for k in range(3)] for _ in range(14)]     # 14 holders x 3 chains
    warm = [rollout.pooled_stream("warmup") for _ in range(14 * 3)]
    assert all(c[k] is chains[0][k] for c in chains for k in range(3)) and len({id(s) for s in chains[0]}) == 3
    assert all(w is warm[0] for w in warm) and warm[0] not in chains[0]
    assert rollout._hoist_stream() is rollout.pooled_stream("hoist")
    assert len(made) == 5
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 1)          # another device: its own streams
    assert rollout.pooled_stream("chain", 0) is not chains[0][0] and len(made) == 6
