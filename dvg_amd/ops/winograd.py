"""Winograd F(4x4,3x3) / F(2x2,3x3) wrappers (winograd.hip + the batched GEMM mode of conv_igemm2.hip): weight transform,
input / output transforms and the hand-over kernels that keep the activation between two Winograd layers out of HBM."""
from __future__ import annotations

import torch

from .._lib import check, lib
from ._core import ACT_LRELU, _dev_f32, _p, _run, _stream, is_nhwc, nhwc_empty
from .conv import packed_row_floats


# ----------------------------------------------------------------------------------
# Winograd F(2x2,3x3) form of the 3x3 block (eval-mode deep layers; dvg_amd/csrc/winograd.hip)
# ----------------------------------------------------------------------------------
def winograd_weight(w: torch.Tensor, m: int = 2) -> torch.Tensor:
    """U = G g G^T of a Conv2d weight (Cout,Cin,3,3) for F(m x m, 3x3), in the k16 layout of the (m+2)^2 batched GEMMs:
    logical shape ((m+2)^2, Cin/16, 1, Cout, row), in memory [(m+2)^2][Cout/64][Cin/16][64][row]."""
    _dev_f32(w, "winograd_weight")
    w = w.detach().contiguous()
    co, ci, kh, kw = w.shape
    if (kh, kw) != (3, 3) or ci % 16 or m not in (2, 4):
        raise RuntimeError("winograd_weight: (Cout, Cin % 16 == 0, 3, 3) and m in (2, 4) expected")
    u = torch.empty(((m + 2) ** 2, ci // 16, 1, co, packed_row_floats()), device=w.device, dtype=torch.float32)
    check(lib().dvg_winograd_weight(_p(w), _p(u), co, ci, m, _stream()), "winograd_weight")
    return u


def winograd_ok(n, c, h, w, cout, m: int = 2) -> bool:
    """Shapes the Winograd path takes: whole 128-row GEMM tiles and 64-channel blocks."""
    return h % m == 0 and w % m == 0 and c % 64 == 0 and cout % 64 == 0 and (n * (h // m) * (w // m)) % 128 == 0


class WinoV:
    """The Winograd F(4x4,3x3) input transform V (36, T, C) of an activation (N,C,H,W) that was never materialised: what
    conv3x3_winograd(..., to_v=True) hands to the next layer's conv3x3_winograd instead of y."""
    __slots__ = ("v", "shape", "up")

    def __init__(self, v, shape, up=False):
        # up: `shape` is the UPSAMPLED map the transform was taken of (conv3x3_winograd(to_v="up")): only a layer called with
        # upsample=True may consume it
        self.v, self.shape, self.up = v, tuple(shape), bool(up)

    @property
    def device(self):
        return self.v.device

    @property
    def requires_grad(self):
        return False


def winograd_chain_ok(n, c, h, w):
    """Shapes dvg_winograd_output_input takes (the fused output -> input transform between two F(4x4,3x3) layers)."""
    return h == w and h in (8, 16, 32) and c % 64 == 0 and n > 0


def winograd_up_chain_ok(n, c, h, w):
    """Shapes dvg_winograd_output_up_input takes (last layer of a decoder block -> upsample -> first conv of the next)."""
    return h == w and h == 8 and c % 64 == 0 and n > 0


def winograd_pool_chain_ok(n, c, h, w):
    """Shapes dvg_winograd_output_pool_input takes (last layer of an encoder stage -> first layer of the next stage)."""
    return h == w and h in (16, 32) and c % 64 == 0 and n > 0


def conv3x3_winograd(x, u, scale, shift, *, act=ACT_LRELU, slope=0.2, pool=False, return_v=False, to_v=False,
                     upsample=False, addend=None, y_from=0):
    """y = act(conv3x3(x) * scale + shift) (+ pooled y) through input transform -> (m+2)^2 batched GEMMs -> output
    transform; m (2 or 4) follows from u's leading dimension (16 or 36).  x may be a WinoV (the previous layer's to_v=True
    result: no input transform); to_v=True returns the NEXT layer's input transform as a WinoV instead of y (m = 4,
    winograd_chain_ok shapes: dvg_winograd_output_input); pool=True with to_v=True returns (y, WinoV of maxpool2x2(y)): the
    last layer of an encoder stage handing over to the first layer of the next (dvg_winograd_output_pool_input).
    to_v="up": the WinoV is the input transform of nearest_up2(y) (dvg_winograd_output_up_input; the consumer is a layer called
    with upsample=True, which then skips its own input transform).
    upsample (m = 4): x is read through nearest-x2 upsampling (output 2H x 2W).  addend (m = 4, no pool): raw partial sums
    in the output's shape, y = act((conv + addend) * scale + shift) - the hoisted skip half of a decoder block's first conv.
    y_from (pool only): y is stored for the images [y_from, N) only (N - y_from images; None when y_from == N)."""
    from_v = isinstance(x, WinoV)
    if from_v:
        n, c, h, w = x.shape
        if x.up != bool(upsample):
            raise RuntimeError("conv3x3_winograd: a WinoV taken through the upsampling needs upsample=True (and only then)")
    else:
        _dev_f32(x, "conv3x3_winograd.x")
        assert is_nhwc(x), "conv3x3_winograd: x must be NHWC in memory"
        n, c, h, w = x.shape
        if upsample:
            h, w = 2 * h, 2 * w
    cout, npos = u.shape[3], u.shape[0]
    mt = {16: 2, 36: 4}.get(npos, 0)
    if mt == 0 or tuple(u.shape) != (npos, c // 16, 1, cout, packed_row_floats()) or not winograd_ok(n, c, h, w, cout, mt):
        raise RuntimeError(f"conv3x3_winograd: unsupported shape x {tuple(x.shape)} u {tuple(u.shape)}")
    if not 0 <= y_from <= n or (y_from and not pool):      # (all argument checks BEFORE the first launch: ADVICE r05)
        raise RuntimeError("conv3x3_winograd: y_from needs the pooled output and 0 <= y_from <= N")
    if (from_v or to_v or upsample or addend is not None) and (mt != 4 or return_v):
        raise RuntimeError("conv3x3_winograd: WinoV hand-over / upsample / addend need F(4x4,3x3) and no return_v")
    if addend is not None and pool:
        raise RuntimeError("conv3x3_winograd: addend excludes the pooled output")
    if to_v == "up" and (pool or addend is not None or not winograd_up_chain_ok(n, cout, h, w)):
        raise RuntimeError(f"conv3x3_winograd: to_v='up' unsupported for output {(n, cout, h, w)} pool={pool}")
    if addend is not None:
        _dev_f32(addend, "conv3x3_winograd.addend")
        if tuple(addend.shape) != (n, cout, h, w) or not is_nhwc(addend):
            raise RuntimeError(f"conv3x3_winograd: addend {tuple(addend.shape)} must be NHWC {(n, cout, h, w)}")
    if to_v and to_v != "up" and not (winograd_pool_chain_ok(n, cout, h, w) if pool else winograd_chain_ok(n, cout, h, w)):
        raise RuntimeError(f"conv3x3_winograd: to_v unsupported for output {(n, cout, h, w)} pool={pool}")
    t = n * (h // mt) * (w // mt)
    dev = x.device
    m = torch.empty((npos, t, cout), device=dev, dtype=torch.float32)
    if from_v:
        v = x.v
        if tuple(v.shape) != (npos, t, c):
            raise RuntimeError("conv3x3_winograd: WinoV does not match its shape")
    else:
        v = torch.empty((npos, t, c), device=dev, dtype=torch.float32)
        _run("winograd_input", 0.0, 4.0 * (x.numel() + v.numel()), lib().dvg_winograd_input, _p(x), _p(v), n, h, w, c, mt,
             int(upsample), _stream())
    _run("winograd_gemm", 2.0 * npos * t * c * cout, 4.0 * (v.numel() + m.numel() + u.numel()), lib().dvg_gemm_batched_k16,
         _p(v), _p(u), _p(m), npos, t // 16, 16, c, cout, _stream(), alg_flops=2.0 * n * h * w * cout * 9 * c)
    if to_v and pool:
        y = nhwc_empty(n - y_from, cout, h, w, dev) if y_from < n else None
        vn = torch.empty((npos, t // 4, cout), device=dev, dtype=torch.float32)
        _run("winograd_output_pool_input", 0.0, 4.0 * (m.numel() + (0 if y is None else y.numel()) + vn.numel()),
             lib().dvg_winograd_output_pool_input, _p(m), _p(scale), _p(shift), _p(y), _p(vn), n, h, w, cout, act, slope, y_from,
             _stream())
        return y, WinoV(vn, (n, cout, h // 2, w // 2))
    if to_v == "up":
        vn = torch.empty((npos, 4 * t, cout), device=dev, dtype=torch.float32)
        _run("winograd_output_up_input", 0.0, 4.0 * (m.numel() + vn.numel()), lib().dvg_winograd_output_up_input, _p(m), _p(scale),
             _p(shift), _p(vn), n, h, w, cout, act, slope, _stream())
        return WinoV(vn, (n, cout, 2 * h, 2 * w), up=True)
    if to_v:
        vn = torch.empty((npos, t, cout), device=dev, dtype=torch.float32)
        _run("winograd_output_input", 0.0, 4.0 * (m.numel() + vn.numel()), lib().dvg_winograd_output_input, _p(m), _p(scale),
             _p(shift), _p(vn), n, h, w, cout, act, slope, _p(addend), _stream())
        return WinoV(vn, (n, cout, h, w))
    y = nhwc_empty(n - y_from, cout, h, w, dev) if y_from < n else None
    yp = nhwc_empty(n, cout, h // 2, w // 2, dev) if pool else None
    _run("winograd_output", 0.0, 4.0 * (m.numel() + (0 if y is None else y.numel()) + (0 if yp is None else yp.numel())),
         lib().dvg_winograd_output, _p(m), _p(scale), _p(shift), _p(y), _p(yp), n, h, w, cout, act, slope, mt, _p(addend), y_from,
         _stream())
    if return_v:   # the input transform (P, T, C): the Winograd-form weight gradient's second operand (training)
        return ((y, yp) if pool else y), v
    return (y, yp) if pool else y


def stem_up_winograd_input(vec, w_kn, k, scale, shift, cout, *, act=ACT_LRELU, slope=0.2):
    """Decoder stem + BN + activation, nearest-x2 upsampling and the F(4x4,3x3) input transform of the result in one launch
    (dvg_stem_up_winograd_input): a WinoV (.up) of shape (M, cout, 8, 8) for the x half of the first decoder block's concat
    conv; the 4 x 4 map is never written.  w_kn as stem_gemm takes it."""
    _dev_f32(vec, "stem_up_winograd_input.vec")
    if vec.dim() != 2 or vec.stride(1) != 1:
        vec = vec.contiguous().view(vec.shape[0], -1)
    m = vec.shape[0]
    kp, n = w_kn.shape
    if vec.shape[1] != k or n != 16 * cout or cout % 16:
        raise RuntimeError("stem_up_winograd_input: shape mismatch")
    v = torch.empty((36, 4 * m, cout), device=vec.device, dtype=torch.float32)
    _run("stem_up_winograd_input", 2.0 * m * n * k, 4.0 * (m * k + n * k + v.numel()), lib().dvg_stem_up_winograd_input, _p(vec),
         vec.stride(0), _p(w_kn), kp, _p(scale), _p(shift), _p(v), m, cout, k, act, slope, _stream())
    return WinoV(v, (m, cout, 8, 8), up=True)
