"""Encoder / decoder backbones of the DVG frame predictor, built from stage plans.

The classes reproduce the reference's *attribute tree* (c1..c5/c6, upc1..upc5/upc6,
`.main` Sequentials holding nn.Conv2d / nn.BatchNorm2d parameter containers) so that
state_dict keys, `encoder.apply(utils.init_weights)` (class-name matching, utils.py:304-311)
and reference-pickled checkpoints (train.py:380-383) keep working — SURVEY.md §8(b).
None of the torch.nn compute modules is ever *called*: every forward goes through
dvg_amd.fused -> libdvg_hip.so.  Activations and the returned skip tensors are
(N,C,H,W)-shaped with channels_last (NHWC) strides.

Reference behaviour per class is cited at the class.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import fused, ops
from ..ops import ACT_SIGMOID, ACT_TANH

# channel plans per stage -------------------------------------------------------------
# vgg: list of stages, each a list of (nin, nout); "nc" stands for the image channels.
_VGG_ENC = {
    64: [[("nc", 64), (64, 64)], [(64, 128), (128, 128)], [(128, 256), (256, 256), (256, 256)],
         [(256, 512), (512, 512), (512, 512)]],
    128: [[("nc", 64), (64, 64)], [(64, 128), (128, 128)], [(128, 256), (256, 256), (256, 256)],
          [(256, 512), (512, 512), (512, 512)], [(512, 512), (512, 512), (512, 512)]],
}
_VGG_DEC = {
    64: [[(1024, 512), (512, 512), (512, 256)], [(512, 256), (256, 256), (256, 128)], [(256, 128), (128, 64)],
         [(128, 64)]],
    128: [[(1024, 512), (512, 512), (512, 512)], [(1024, 512), (512, 512), (512, 256)],
          [(512, 256), (256, 256), (256, 128)], [(256, 128), (128, 64)], [(128, 64)]],
}
_DCGAN_ENC = {64: [("nc", 64), (64, 128), (128, 256), (256, 512)],
              128: [("nc", 64), (64, 128), (128, 256), (256, 512), (512, 512)]}
_DCGAN_DEC = {64: [(1024, 256), (512, 128), (256, 64)], 128: [(1024, 512), (1024, 256), (512, 128), (256, 64)]}


def _skip_nhwc(s):
    """A skip operand as the kernels take it: NHWC in memory; an ops.SharedBlocks (the time-batched decoder calls of
    train.py:227-231 share skip tensors across calls) keeps its sharing pattern."""
    if isinstance(s, ops.SharedBlocks):
        return s.like(ops.to_nhwc(s.t))
    return ops.to_nhwc(s)


def _frame(x: torch.Tensor) -> torch.Tensor:
    if x.dim() != 4:
        raise RuntimeError(f"expected a (B,C,H,W) frame batch, got {tuple(x.shape)}")
    return x


class vgg_layer(nn.Module):
    """Conv2d(nin,nout,3,1,1) + BatchNorm2d + LeakyReLU(0.2)   [vgg_64.py:5-15]"""

    def __init__(self, nin, nout):
        super().__init__()
        self.main = nn.Sequential(nn.Conv2d(nin, nout, 3, 1, 1), nn.BatchNorm2d(nout), nn.LeakyReLU(0.2, inplace=True))

    def forward(self, input):
        conv, bn = self.main[0], self.main[1]
        if conv.in_channels % 32:
            return fused.conv3_first_bn_act(conv, bn, _frame(input))
        return fused.conv3_bn_act(conv, bn, ops.to_nhwc(input))


class dcgan_conv(nn.Module):
    """Conv2d(nin,nout,4,2,1) + BatchNorm2d + LeakyReLU(0.2)   [dcgan_64.py:4-14]"""

    def __init__(self, nin, nout):
        super().__init__()
        self.main = nn.Sequential(nn.Conv2d(nin, nout, 4, 2, 1), nn.BatchNorm2d(nout), nn.LeakyReLU(0.2, inplace=True))

    def forward(self, input):
        conv, bn = self.main[0], self.main[1]
        if conv.in_channels % 32:
            return fused.conv4s2_first_bn_act(conv, bn, _frame(input))
        return fused.conv4s2_bn_act(conv, bn, ops.to_nhwc(input))


class dcgan_upconv(nn.Module):
    """ConvTranspose2d(nin,nout,4,2,1) + BatchNorm2d + LeakyReLU(0.2)   [dcgan_64.py:16-26]"""

    def __init__(self, nin, nout):
        super().__init__()
        self.main = nn.Sequential(nn.ConvTranspose2d(nin, nout, 4, 2, 1), nn.BatchNorm2d(nout),
                                  nn.LeakyReLU(0.2, inplace=True))

    def forward(self, input, skip=None):
        # `input` may be the already concatenated tensor (reference call style) or the pair is
        # passed separately by our decoder so that the concat is never materialised.
        return fused.convT4s2_bn_act(self.main[0], self.main[1], ops.to_nhwc(input),
                                     None if skip is None else ops.to_nhwc(skip))


def _head(dim):
    # Conv2d(512,dim,4,1,0) + BatchNorm2d + Tanh   [vgg_64.py:44-48, dcgan_64.py:42-46]
    return nn.Sequential(nn.Conv2d(512, dim, 4, 1, 0), nn.BatchNorm2d(dim), nn.Tanh())


def _stem(dim):
    # ConvTranspose2d(dim,512,4,1,0) + BatchNorm2d + LeakyReLU   [vgg_64.py:65-69, dcgan_64.py:62-67]
    return nn.Sequential(nn.ConvTranspose2d(dim, 512, 4, 1, 0), nn.BatchNorm2d(512), nn.LeakyReLU(0.2, inplace=True))


# --------------------------------------------------------------------------------------
# VGG
# --------------------------------------------------------------------------------------
class VggEncoder(nn.Module):
    """vgg_64.encoder (vgg_64.py:17-57) / vgg_128.encoder (vgg_128.py:16-63).
    forward(x) -> (h.view(-1,dim), [skip per stage])."""
    RES = 64

    def __init__(self, dim, nc=1):
        super().__init__()
        self.dim = dim
        plan = _VGG_ENC[self.RES]
        for s, stage in enumerate(plan, start=1):
            setattr(self, f"c{s}", nn.Sequential(*[vgg_layer(nc if a == "nc" else a, b) for a, b in stage]))
        setattr(self, f"c{len(plan) + 1}", _head(dim))
        self.mp = nn.MaxPool2d(kernel_size=2, stride=2, padding=0)

    def _stages(self):
        n = 1
        while hasattr(self, f"c{n + 1}"):
            n += 1
        return [getattr(self, f"c{s}") for s in range(1, n)], getattr(self, f"c{n}")

    def features(self, input, skips_from=0):
        """(head output, skip tensors).  skips_from = k (eval mode, no grad): the skip tensors are wanted for the images
        [k, N) only - they come back with N - k images (None each when k == N) and the kernels that can do not store the
        rest: a rollout keeps the skips of the last conditioning frame and discards those of every predicted frame
        (generate_frames.py:154-157), and the full-resolution stage outputs are 126 MB per 64 frames."""
        stages, head = self._stages()
        skips = []
        h = _frame(input)
        k = int(skips_from)
        if k and (self.training or torch.is_grad_enabled()):
            raise RuntimeError("encoder: skips_from is an eval-mode, no-grad option")
        for si, stage in enumerate(stages):
            layers = list(stage)
            if si == 0 and len(layers) == 2 and fused.first_pair_applies(layers[0].main[0], layers[0].main[1],
                                                                         layers[1].main[0], layers[1].main[1], h):
                # eval mode, one input channel: both layers of c1 in one launch (the 64-channel map between them stays on chip)
                full, h = fused.conv3_first_pair(layers[0].main[0], layers[0].main[1], layers[1].main[0], layers[1].main[1], h,
                                                 pool=True, y_from=k)
                skips.append(full)
                continue
            for li, layer in enumerate(layers):
                conv, bn = layer.main[0], layer.main[1]
                last = li == len(layers) - 1
                if conv.in_channels % 32:
                    h = fused.conv3_first_bn_act(conv, bn, h)
                elif last:  # stage output = skip tensor; its 2x2 max-pool feeds the next stage and nothing else (eval: the
                    # pooled map may be handed over as the next stage's Winograd input transform, an ops.WinoV)
                    nxt = stages[si + 1][0].main[0] if si + 1 < len(stages) else None
                    full, h = fused.conv3_bn_act(conv, bn, h, pool=True, next_conv=nxt, y_from=k)
                    skips.append(full)
                else:       # inner layer: its output has one consumer, the next layer (eval: may hand over an ops.WinoV)
                    h = fused.conv3_bn_act(conv, bn, h, next_conv=layers[li + 1].main[0])
        return fused.head_bn_tanh(head[0], head[1], h), skips

    def forward(self, input):
        h, skips = self.features(input)
        return h.view(-1, self.dim), skips

    def encode(self, input, skips_from=0):
        """forward() with the skip tensors of the images [skips_from, N) only (see features)."""
        h, skips = self.features(input, skips_from)
        return h.view(-1, self.dim), skips


class VggDecoder(nn.Module):
    """vgg_64.decoder (vgg_64.py:60-106) / vgg_128.decoder (vgg_128.py:66-120).
    forward([vec, skips]) -> (B,nc,H,W) in (0,1)."""
    RES = 64

    def __init__(self, dim, nc=1):
        super().__init__()
        self.dim = dim
        plan = _VGG_DEC[self.RES]
        self.upc1 = _stem(dim)
        for s, stage in enumerate(plan, start=2):
            mods = [vgg_layer(a, b) for a, b in stage]
            if s == len(plan) + 1:
                mods += [nn.ConvTranspose2d(64, nc, 3, 1, 1), nn.Sigmoid()]
            setattr(self, f"upc{s}", nn.Sequential(*mods))
        self.up = nn.UpsamplingNearest2d(scale_factor=2)

    def forward(self, input):
        vec, skip = input
        n = 1
        while hasattr(self, f"upc{n + 1}"):
            n += 1
        sks = {s: _skip_nhwc(skip[n - s]) for s in range(2, n + 1)}
        fl = self.upc2[0]
        d = fused.stem_bn_act(self.upc1[0], self.upc1[1], vec,
                              next_up=(fl.main[0], fl.main[1], sks[2]) if isinstance(fl, vgg_layer) and n > 2 else None)
        for s in range(2, n + 1):
            sk = sks[s]
            first = True
            mods = list(getattr(self, f"upc{s}"))
            for li, layer in enumerate(mods):
                if isinstance(layer, vgg_layer):
                    # the layer after this one when it is a vgg_layer of the same block (eval: may take an ops.WinoV) ...
                    nxt = mods[li + 1] if li + 1 < len(mods) and isinstance(mods[li + 1], vgg_layer) and s < n else None
                    # ... or, for a block's last layer, the first layer of the NEXT block, which reads this output through `up`
                    # and cat(skip) and nothing else does (vgg_64.py:98-105; eval rollouts: an ops.WinoV through the upsampling)
                    nup = None
                    if nxt is None and li == len(mods) - 1 and s < n:
                        fl = getattr(self, f"upc{s + 1}")[0]
                        nup = (fl.main[0], fl.main[1], sks[s + 1])
                    if first:  # nearest x2 + cat(skip) fused into the tile loader (eval rollouts: x half in Winograd form)
                        d = fused.conv3_bn_act(layer.main[0], layer.main[1], d, sk, upsample=True,
                                               next_conv=None if nxt is None else nxt.main[0], next_up=nup)
                        first = False
                    else:      # an inner layer followed by another vgg_layer of the block may hand over an ops.WinoV
                        d = fused.conv3_bn_act(layer.main[0], layer.main[1], d,
                                               next_conv=None if nxt is None else nxt.main[0], next_up=nup)
                elif isinstance(layer, nn.ConvTranspose2d):
                    d = fused.convT3_last(layer, d, act=ACT_SIGMOID)
        return d

    @torch.no_grad()
    def precompute_frozen_skips(self, skip) -> None:
        """Eval-mode rollouts: the skip halves of every block's concat conv for skip tensors that will not change any more
        (fused.precompute_skip_half); forward() then runs only the x halves."""
        n = 1
        while hasattr(self, f"upc{n + 1}"):
            n += 1
        for s in range(2, n + 1):
            first = next(l for l in getattr(self, f"upc{s}") if isinstance(l, vgg_layer))
            fused.precompute_skip_half(first.main[0], ops.to_nhwc(skip[n - s]), "conv3")


class VggGaussianEncoder(VggEncoder):
    """vgg_64.gaussian_encoder (vgg_64.py:108-159): encoder trunk + mu/logvar heads +
    reparameterisation with eps ~ N(0,1) from the global torch RNG."""

    def __init__(self, dim, output_size, nc=1):
        super().__init__(dim, nc)
        self.output_size = output_size
        self.mu_net = nn.Linear(dim, output_size)
        self.logvar_net = nn.Linear(dim, output_size)

    def reparameterize(self, mu, logvar):
        from .lstm import reparameterize
        return reparameterize(mu, logvar)

    def forward(self, input):
        from .lstm import linear
        h, skips = self.features(input)
        h = h.view(-1, self.dim)
        mu = linear(self.mu_net, h)
        logvar = linear(self.logvar_net, h)
        return self.reparameterize(mu, logvar), mu, logvar, skips


# --------------------------------------------------------------------------------------
# DCGAN
# --------------------------------------------------------------------------------------
class DcganEncoder(nn.Module):
    """dcgan_64.encoder (dcgan_64.py:28-54) / dcgan_128.encoder (dcgan_128.py:28-57)."""
    RES = 64

    def __init__(self, dim, nc=1):
        super().__init__()
        self.dim = dim
        plan = _DCGAN_ENC[self.RES]
        for s, (a, b) in enumerate(plan, start=1):
            setattr(self, f"c{s}", dcgan_conv(nc if a == "nc" else a, b))
        setattr(self, f"c{len(plan) + 1}", _head(dim))

    def forward(self, input):
        return self.encode(input)

    def encode(self, input, skips_from=0):
        """forward() with the skip tensors of the images [skips_from, N) only (same convention as VggEncoder.encode; here every
        skip is also the next layer's input, so nothing is saved - the skips are views)."""
        n = 1
        while hasattr(self, f"c{n + 1}"):
            n += 1
        h = _frame(input)
        k = int(skips_from)
        skips = []
        for s in range(1, n):
            layer = getattr(self, f"c{s}")
            conv, bn = layer.main[0], layer.main[1]
            h = fused.conv4s2_first_bn_act(conv, bn, h) if conv.in_channels % 32 else fused.conv4s2_bn_act(conv, bn, h)
            skips.append(h if not k else (h[k:] if k < h.shape[0] else None))
        head = getattr(self, f"c{n}")
        return fused.head_bn_tanh(head[0], head[1], h).view(-1, self.dim), skips


class DcganDecoder(nn.Module):
    """dcgan_64.decoder (dcgan_64.py:57-88; final Tanh) / dcgan_128.decoder
    (dcgan_128.py:60-94; final Sigmoid)."""
    RES = 64
    FINAL_ACT = ACT_TANH

    def __init__(self, dim, nc=1):
        super().__init__()
        self.dim = dim
        plan = _DCGAN_DEC[self.RES]
        self.upc1 = _stem(dim)
        for s, (a, b) in enumerate(plan, start=2):
            setattr(self, f"upc{s}", dcgan_upconv(a, b))
        final_act = nn.Tanh() if self.FINAL_ACT == ACT_TANH else nn.Sigmoid()
        setattr(self, f"upc{len(plan) + 2}", nn.Sequential(nn.ConvTranspose2d(128, nc, 4, 2, 1), final_act))

    def forward(self, input):
        vec, skip = input
        n = 1
        while hasattr(self, f"upc{n + 1}"):
            n += 1
        d = fused.stem_bn_act(self.upc1[0], self.upc1[1], vec)
        for s in range(2, n):
            layer = getattr(self, f"upc{s}")
            d = fused.convT4s2_bn_act(layer.main[0], layer.main[1], d, _skip_nhwc(skip[n - s]))
        last = getattr(self, f"upc{n}")
        act = ACT_SIGMOID if isinstance(last[1], nn.Sigmoid) else ACT_TANH
        return fused.convT4s2_last(last[0], d, _skip_nhwc(skip[0]), act=act)

    @torch.no_grad()
    def precompute_frozen_skips(self, skip) -> None:
        """Eval-mode rollouts: skip halves of the concat blocks and the skip's share of the last layer's projection for
        skip tensors that will not change any more (see VggDecoder.precompute_frozen_skips)."""
        n = 1
        while hasattr(self, f"upc{n + 1}"):
            n += 1
        for s in range(2, n):
            fused.precompute_skip_half(getattr(self, f"upc{s}").main[0], ops.to_nhwc(skip[n - s]), "convT4s2")
        ops.precompute_skip_proj(ops.to_nhwc(skip[0]), getattr(self, f"upc{n}")[0].weight, 4)
