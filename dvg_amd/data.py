"""Synthetic stand-ins for the datasets of the reference (no network, no torchvision here).

`SyntheticMovingMNIST` follows the trajectory logic of data/moving_mnist.py:38-91 exactly —
`num_digits` 32x32 sprites on a 64x64 canvas, start ~ randint(32), velocity ~ randint(-4,5),
the non-deterministic bounce rules (:56-84), additive compositing clipped at 1 (:90) — but the
sprites come from a seeded in-repo generator instead of MNIST.  Frames are (T,H,W,1) float32 in
[0,1]; a batch is (B,T,H,W,1), which `utils.normalize_data` turns into T x (B,1,H,W)."""
from __future__ import annotations

import sys

import numpy as np
import torch

from .utils import normalize_data


def _sprites(rng: np.random.Generator, n: int, size: int = 32) -> np.ndarray:
    """Digit-like blobs: a few thick random strokes, smoothed, values in [0,1]."""
    out = np.zeros((n, size, size), np.float32)
    yy, xx = np.mgrid[0:size, 0:size]
    for i in range(n):
        img = np.zeros((size, size), np.float32)
        p = rng.uniform(6, size - 6, 2)
        for _ in range(rng.integers(3, 6)):
            q = np.clip(p + rng.normal(0, 7, 2), 4, size - 5)
            for t in np.linspace(0, 1, 24):
                c = p * (1 - t) + q * t
                img = np.maximum(img, np.exp(-((yy - c[0]) ** 2 + (xx - c[1]) ** 2) / (2 * 1.6 ** 2)))
            p = q
        out[i] = np.clip(img * 1.3, 0, 1)
    return out


class SyntheticMovingMNIST:
    def __init__(self, seq_len=20, num_digits=2, image_size=64, seed=1, n_sprites=64, deterministic=False):
        self.seq_len, self.num_digits, self.image_size = seq_len, num_digits, image_size
        self.digit_size = 32
        self.deterministic = deterministic
        self.rng = np.random.default_rng(seed)
        self.data = _sprites(self.rng, n_sprites, self.digit_size)
        self.N = n_sprites

    def __len__(self):
        return 10000

    def _trajectory(self):
        """One sample's RNG draws in the order of moving_mnist.py:43-85: per digit the sprite index, the start and
        the velocity, then the per-frame bounce rules.  Returns (ids (num_digits,), pos (num_digits, T, 2) = (sy, sx))."""
        rng, S, D = self.rng, self.image_size, self.digit_size
        ids = np.zeros(self.num_digits, np.int32)
        pos = np.zeros((self.num_digits, self.seq_len, 2), np.int32)
        for n in range(self.num_digits):
            ids[n] = rng.integers(self.N)
            sx, sy = int(rng.integers(S - D)), int(rng.integers(S - D))
            dx, dy = int(rng.integers(-4, 5)), int(rng.integers(-4, 5))
            for t in range(self.seq_len):
                if sy < 0:
                    sy = 0
                    if self.deterministic:
                        dy = -dy
                    else:
                        dy, dx = int(rng.integers(1, 5)), int(rng.integers(-4, 5))
                elif sy >= S - D:
                    sy = S - D - 1
                    if self.deterministic:
                        dy = -dy
                    else:
                        dy, dx = int(rng.integers(-4, 0)), int(rng.integers(-4, 5))
                if sx < 0:
                    sx = 0
                    if self.deterministic:
                        dx = -dx
                    else:
                        dx, dy = int(rng.integers(1, 5)), int(rng.integers(-4, 5))
                elif sx >= S - D:
                    sx = S - D - 1
                    if self.deterministic:
                        dx = -dx
                    else:
                        dx, dy = int(rng.integers(-4, 0)), int(rng.integers(-4, 5))
                pos[n, t] = (sy, sx)
                sy += dy
                sx += dx
        return ids, pos

    def __getitem__(self, index):
        S, D = self.image_size, self.digit_size
        ids, pos = self._trajectory()
        x = np.zeros((self.seq_len, S, S, 1), np.float32)
        for n in range(self.num_digits):
            digit = self.data[ids[n]]
            for t in range(self.seq_len):
                sy, sx = pos[n, t]
                x[t, sy:sy + D, sx:sx + D, 0] += digit
        x[x > 1] = 1.0
        return x

    def trajectories(self, batch_size: int):
        """The host half of a batch: sprite ids (B, num_digits) and positions (B, num_digits, T, 2) int32, drawn in the
        order `batch()` draws them (same generator state -> same batch)."""
        trajs = [self._trajectory() for _ in range(batch_size)]
        ids = np.stack([t[0] for t in trajs])
        pos = np.stack([t[1] for t in trajs])
        lim = self.image_size - self.digit_size
        if ids.min() < 0 or ids.max() >= self.N or pos.min() < 0 or pos.max() > lim:
            raise RuntimeError("SyntheticMovingMNIST: trajectory outside the canvas")
        return ids, pos

    def batch_device(self, batch_size: int, device) -> list:
        """The same batch as `utils.normalize_data(opt, dtype, self.batch(batch_size))` - bit for bit, given the same
        generator state - composited on the GPU (dvg_moving_mnist_compose) straight into the T x (B,1,S,S) layout:
        only the integer trajectories (a few KB) cross PCIe."""
        return self.compose_device(*self.trajectories(batch_size), device)

    def compose_device(self, ids, pos, device) -> list:
        """The device half: additive compositing + clip + normalize_data's layout in one kernel."""
        from . import ops
        if getattr(self, "_dev_sprites", None) is None or self._dev_sprites.device != torch.device(device):
            self._dev_sprites = torch.from_numpy(self.data).to(device)
        out = ops.moving_mnist_compose(self._dev_sprites, torch.from_numpy(ids).to(device),
                                       torch.from_numpy(pos).to(device), self.seq_len, self.image_size)
        return [out[t] for t in range(self.seq_len)]

    def batch(self, batch_size: int) -> torch.Tensor:
        return torch.from_numpy(np.stack([self[i] for i in range(batch_size)]))  # (B,T,H,W,1)


def synthetic_video(batch, seq_len, channels, res, seed=1) -> torch.Tensor:
    """(B,T,H,W,C) U[0,1]-textured clips with temporal coherence, for the KTH/BAIR/UCF-shaped configs."""
    rng = np.random.default_rng(seed)
    base = rng.random((batch, 1, res, res, channels), dtype=np.float32)
    # float32 draws, scaled / accumulated / clipped in place: the float64 form took longer on the host than a dcgan_64
    # training iteration at this shape takes on the GPU (train.py draws batches on a background thread)
    drift = rng.standard_normal((batch, seq_len, res, res, channels), dtype=np.float32)
    drift *= np.float32(0.05)
    np.cumsum(drift, axis=1, out=drift)
    drift += base
    return torch.from_numpy(np.clip(drift, 0, 1, out=drift))


def make_batch_generator(opt, seq_len, seed, device=None):
    """Yields `load()` callables: the host half of a batch has been drawn when the callable is yielded, calling it (on the
    thread that owns the GPU stream) puts the batch on the device as normalize_data's list of T x (B,C,H,W) frames.
    smmnist: the host draws the integer trajectories, the device composites them (bit-identical to the host batch).
    `--data_root` is NOT read: there are no dataset files (nor torchvision / network) in this environment.  smmnist is
    the reference's trajectory generator over seeded in-repo sprites; every other dataset name must be acknowledged with
    --synthetic_data, otherwise a reference command line would silently 'train' on noise."""
    if opt.dataset != 'smmnist' and not getattr(opt, 'synthetic_data', False):
        raise SystemExit(f"train.py: no loader for --dataset {opt.dataset} here (--data_root {opt.data_root!r} is not read). "
                         "Pass --synthetic_data to train on synthetic clips of that dataset's shape.")
    if opt.rank == 0:
        what = ("Moving-MNIST trajectories over synthetic sprites (not MNIST digits)" if opt.dataset == 'smmnist'
                else f"random textured clips shaped like {opt.dataset}")
        print(f"WARNING: synthetic data - {what}; --data_root is ignored", file=sys.stderr)
    if opt.dataset == 'smmnist':
        ds = SyntheticMovingMNIST(seq_len=seq_len, num_digits=opt.num_digits, image_size=opt.image_width, seed=seed)
        while True:
            ids, pos = ds.trajectories(opt.local_batch)
            yield lambda ids=ids, pos=pos: ds.compose_device(ids, pos, device or torch.device('cuda'))
    k = 0
    while True:
        seq = synthetic_video(opt.local_batch, seq_len, opt.channels, opt.image_width, seed=seed + k)
        yield lambda seq=seq: normalize_data(opt, torch.cuda.FloatTensor, seq)[0]
        k += 1
