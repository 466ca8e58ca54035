"""How train.py replays its iterations: one hipGraph per iteration (GraphedIteration), a chain of hipGraphs cut at the gradient
all-reduces for several ranks (SegmentedIteration), and the background thread that draws the next batch while the GPU works
(BatchPrefetcher).  Split from train.py in r06 (re-exported there: `train.GraphedIteration` etc. keep working); the closures
and the Trainer state they drive stay in train.py."""
import sys

import torch

from . import fused

class GraphedIteration:
    """`Trainer.iteration` as ONE hipGraph (torch.cuda.CUDAGraph): forward, the three backward passes, the gradient
    all-reduces and the four fused Adam steps - several thousand launches whose Python + ctypes cost otherwise bounds
    the small per-GPU batches of the data-parallel configs (dcgan_64 at 16 / GPU: 41 ms of kernels in a 132 ms
    iteration).  The first `warmup` calls run eagerly (they are ordinary training steps: weight packs, LDS attributes,
    GP prior initialisation and allocator warm-up happen there); the next call captures and replays; later calls replay.
    After each replay the version counters of all parameters and buffers are bumped (the graph writes them through raw
    pointers; the packed-weight / BN-fold caches key on versions) and the optimisers' host-side step counts advance.
    A change of any learning rate (MultiStepLR, train.py:105-106) triggers a re-capture."""

    def __init__(self, trainer, warmup: int = 2):
        self.tr, self.warmup = trainer, warmup
        self.calls = 0
        self.graph = None
        self.sig = None
        self.failed = False          # a capture raised: eager iterations from then on
        self.outs = self._keepalive = None

    def _replay(self):
        self.graph.replay()

    def _signature(self, x):
        lrs = tuple(g['lr'] for o in self.tr.optimizers() for g in o.param_groups)
        return lrs, tuple(tuple(t.shape) for t in x), self.tr.opt.ft, tuple(m.training for m in self.tr.modules)

    def _release(self):
        """Drop the captured graph(s) and everything they keep alive BEFORE a re-capture allocates a new private pool
        (a MultiStepLR milestone re-captures: two pools of 12-17 GB of saved Winograd transforms need not coexist)."""
        self.graph = None
        self.outs = None
        self._keepalive = None
        torch.cuda.synchronize()

    def _capture(self, x):
        from dvg_amd.rollout import snapshot_eager_caches
        tr = self.tr
        self._release()
        self.static_x = [t.clone() for t in x]
        for o in tr.optimizers():
            o.begin_capture()
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self._capture_body(self.static_x)
        for o in tr.optimizers():
            o.end_capture()                          # every group a captured zero_grads() ticked was stepped in the capture
        self._keepalive = snapshot_eager_caches()    # eager tensors the graph reads by raw pointer stay alive with it
        self.sig = self._signature(x)

    def _capture_body(self, static_x):
        tr = self.tr
        with torch.cuda.graph(self.graph, capture_error_mode="thread_local"):   # see rollout.CAPTURE_KW
            mse_latent, loss = tr._train_model_dev(self.static_x)
            fp = gp = None
            if tr.opt.ft:
                fp, gp = tr._finetune_dev(self.static_x)
            self.outs = (mse_latent, loss, fp, gp)

    def __call__(self, x):
        tr = self.tr
        self.calls += 1
        if self.calls <= self.warmup or self.failed or fused.sync_bn_world() > 1:   # sync-BN: collectives inside every pass
            return tr.iteration(x)
        if self.graph is None or self.sig != self._signature(x):
            try:
                self._capture(x)
            except Exception as e:   # noqa: BLE001 - out of memory for the pool, a partially used Adam group, ...
                # A capture executes nothing: parameters, moments and BatchNorm statistics are untouched.  Fall back to the
                # eager iteration for the rest of the run (no re-exec, no retry: the same capture would fail again).
                self.failed = True
                self._release()
                for o in tr.optimizers():
                    o._captured_groups = []
                tr._segmenter = None
                # ... except the module-level caches: a miss DURING the capture stored a tensor from the graph's pool whose
                # fill was recorded, never run, under the current parameter version.  The eager iteration below must not
                # hit those entries (never-written, freed memory): drop every version-keyed cache, rebuilt on demand.
                from dvg_amd.rollout import drop_version_keyed_caches
                drop_version_keyed_caches()
                tr._ft_cache = None
                tr.frame_predictor.hidden = None
                print(f"train: hipGraph capture failed ({type(e).__name__}: {str(e)[:200]}); continuing with eager "
                      "iterations", file=sys.stderr, flush=True)
                return tr.iteration(x)
        for o in tr.optimizers():
            o.check_graph_fresh()        # an eager step() since the capture left the device-side Adam step counts behind
        for dst, src in zip(self.static_x, x):
            dst.copy_(src)
        self._replay()
        tr._iters = getattr(tr, '_iters', 0) + 1
        for o in tr.optimizers():
            o.after_graph_replay()
        with torch.no_grad():
            for m in tr.modules:
                for b in m.buffers():
                    torch.autograd.graph.increment_version(b)   # BatchNorm running statistics
        mse_latent, loss, fp, gp = self.outs
        T = tr.opt.n_past + tr.opt.n_future
        v = float(mse_latent) / T
        tr.last_loss = float(loss)
        temp = (float(fp) + float(gp)) / T if tr.opt.ft else 0
        return v, v, temp


class SegmentedIteration(GraphedIteration):
    """The data-parallel form of GraphedIteration: the iteration is captured as a CHAIN of hipGraphs cut at the gradient
    all-reduces, which run eagerly between the segments (train_model: [forward + decoder / LSTM / GP backward] -> start
    all-reduce of their ranges -> [encoder backward] -> start the encoder range, wait for both -> [four Adam steps +
    LSTM closure] -> all-reduce -> [LSTM Adam + GP closure] -> all-reduce -> [GP Adam]).  No collective is ever inside a
    captured graph - RCCL runs them on its own stream exactly as in the eager loop, overlap with the encoder phase
    included - and the ≈2 500 launches of an iteration still replay without Python.  All segments share one memory pool
    and are always replayed in capture order."""

    def _release(self):
        self.items, self._pool, self._cur, self._ctx = [], None, None, None
        super()._release()

    def _capture(self, x):
        from dvg_amd.rollout import snapshot_eager_caches
        tr = self.tr
        self._release()
        self.static_x = [t.clone() for t in x]
        for o in tr.optimizers():
            o.begin_capture()
        torch.cuda.synchronize()
        seg = self
        tr._segmenter = seg
        try:
            seg._begin()
            mse_latent, loss = tr._train_model_dev(self.static_x)
            fp = gp = None
            if tr.opt.ft:
                fp, gp = tr._finetune_dev(self.static_x)
            self.outs = (mse_latent, loss, fp, gp)
            seg._end()
        except BaseException:
            if self._ctx is not None:          # a segment is still being captured: end the capture before unwinding
                ctx, self._ctx, self._cur = self._ctx, None, None
                try:
                    ctx.__exit__(*sys.exc_info())
                except Exception:   # noqa: BLE001 - the original exception is the one to report
                    pass
            raise
        finally:
            tr._segmenter = None
        for o in tr.optimizers():
            o.end_capture()
        self.graph = [g for kind, g in self.items if kind == "graph"]   # (truthy: "captured"; replay goes through items)
        self._keepalive = snapshot_eager_caches()
        self.sig = self._signature(x)

    def _begin(self):
        self._cur = torch.cuda.CUDAGraph()
        # thread_local: calls made by OTHER threads while a segment is being captured (the c10d watchdog polling the events
        # of earlier eager collectives) are none of the capture's business; the autograd engine's worker threads still
        # launch into the capturing stream
        kw = {"capture_error_mode": "thread_local"}
        if self._pool is not None:
            kw["pool"] = self._pool
        self._ctx = torch.cuda.graph(self._cur, **kw)
        self._ctx.__enter__()

    def _end(self):
        self._ctx.__exit__(None, None, None)
        if self._pool is None:
            self._pool = self._cur.pool()
        self.items.append(("graph", self._cur))
        self._cur = self._ctx = None

    def cut(self, actions):
        self._end()
        self.items.append(("eager", actions))
        self._begin()

    def _replay(self):
        for kind, item in self.items:
            if kind == "graph":
                item.replay()
            else:
                self.tr._run_ar(item)

    @property
    def n_segments(self):
        return sum(1 for k, _ in self.items if k == "graph")


class BatchPrefetcher:
    """The host half of the input pipeline on a background thread, `depth` batches ahead.  A training iteration ends with
    the host waiting for the GPU (the closures' loss values are read back like train.py:361-362 does); drawing the next
    batch only then left the GPU idle for the whole host-side generation (19 ms of a 102 ms dcgan_64 iteration).  The
    thread runs while the main thread waits (the wait releases the GIL); batches come out in the generator's order."""

    def __init__(self, gen, depth=2):
        import queue
        import threading
        self.gen, self.q = gen, queue.Queue(depth)
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()

    _END = object()

    def _run(self):
        try:
            for item in self.gen:
                self.q.put(item)
            self.q.put(self._END)
        except BaseException as e:   # noqa: BLE001 - handed to the consumer
            self.q.put(e)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.q.get()
        if item is self._END:
            self.q.put(item)         # stay exhausted
            raise StopIteration
        if isinstance(item, BaseException):
            raise item
        return item
